// capi_device.hip — device-side entry points of include/gdpt.h: scene upload (own BVH2 build + HBM
// layout), the five-buffer render, gradient assembly, the Poisson solve and the whole GradPath pipeline.
#include "../../../include/gdpt.h"
#include "../capi_common.h"
#include "../device_scene.h"
#include "../host/bvh.h"
#include "../host/tri_precompute.h"
#include "poisson_kernels.h"
#include "render_kernels.h"
#include "scene_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

namespace {

using gdpt::ck;

template <class T>
T *upload(const std::vector<T> &v) {
    if (v.empty()) return nullptr;
    T *d = nullptr;
    ck(hipMalloc((void **)&d, v.size() * sizeof(T)), "hipMalloc(scene)");
    ck(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(scene)");
    return d;
}

} // namespace

static constexpr int kWavefrontDefault = 0;       // HBM scenes: 1 = wavefront pipeline by default, 0 = lane machine
static constexpr double kSbvhBudget = 1.0;       // extra references / primitives the spatial-split build may add (it adds ~0.1-0.35; knob sbvh overrides)
static constexpr double kPresplitBudget = 0.0;   // extra references / primitives (GDPT_PRESPLIT overrides)

namespace gdpt {

void build_scene(const GdptSceneDesc *desc, int device, GdptScene *sc) {
    if (!desc) throw std::runtime_error("gdpt_scene_upload: null scene description");
    int ndev = 0;
    ck(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev <= 0) throw std::runtime_error("gdpt_scene_upload: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) throw std::runtime_error("gdpt_scene_upload: bad device index");
    ck(hipSetDevice(device), "hipSetDevice");
    sc->device = device;

    const GdptCamera &cam = desc->camera;
    if (cam.width <= 0 || cam.height <= 0) throw std::runtime_error("gdpt_scene_upload: empty film");
    for (int m = 0; m < desc->num_materials; m++) {
        int t = desc->materials[m].type;
        if (t < 0 || t > GDPT_MAT_DISNEY_BSDF) throw std::runtime_error("gdpt_scene_upload: unknown material type");
        for (int k = 0; k < GDPT_MAT_MAX_TEX; k++) {
            const GdptTexture &tx = desc->materials[m].tex[k];
            if (tx.type == GDPT_TEX_IMAGE && (tx.image_id < 0 || tx.image_id >= desc->num_images))
                throw std::runtime_error("gdpt_scene_upload: texture references a missing image");
        }
    }

    // ---- flatten primitives: triangles in (shape, triangle) order = global id; spheres after them ----
    std::vector<DevTriShade> tris;
    std::vector<DevSphere> spheres;
    std::vector<DevPrim> prim_in;          // input order (gid order, spheres last)
    std::vector<gdpt::PrimBounds> bounds;
    std::vector<float> tri_verts;          // the fp32 triangles the intersection test sees (9 floats each), for presplit
    float lb[3], ub[3];
    for (int k = 0; k < 3; k++) { lb[k] = std::numeric_limits<float>::infinity(); ub[k] = -lb[k]; }
    for (int s = 0; s < desc->num_shapes; s++) {
        const GdptShape &sh = desc->shapes[s];
        if (sh.material_id < 0 || sh.material_id >= desc->num_materials) throw std::runtime_error("gdpt_scene_upload: shape without a valid material");
        if (sh.area_light_id >= desc->num_lights) throw std::runtime_error("gdpt_scene_upload: bad area light id");
        if (sh.type != GDPT_SHAPE_TRIMESH) continue;
        if (!sh.positions || !sh.indices) throw std::runtime_error("gdpt_scene_upload: mesh without positions/indices");
        for (int i = 0; i < sh.num_vertices; i++)
            for (int k = 0; k < 3; k++) { float p = (float)sh.positions[3 * i + k]; lb[k] = std::min(lb[k], p); ub[k] = std::max(ub[k], p); }
        for (int t = 0; t < sh.num_triangles; t++) {
            DevTriShade ts{};
            DevPrim pr{};
            gdpt::PrimBounds pb;
            float v[3][3];
            double pos64[3][3];
            for (int k = 0; k < 3; k++) { pb.bmin[k] = std::numeric_limits<float>::infinity(); pb.bmax[k] = -pb.bmin[k]; }
            for (int i = 0; i < 3; i++) {
                int vi = sh.indices[3 * t + i];
                if (vi < 0 || vi >= sh.num_vertices) throw std::runtime_error("gdpt_scene_upload: mesh index out of range");
                for (int k = 0; k < 3; k++) {
                    pos64[i][k] = sh.positions[3 * vi + k];
                    v[i][k] = (float)sh.positions[3 * vi + k];
                    pb.bmin[k] = std::min(pb.bmin[k], v[i][k]); pb.bmax[k] = std::max(pb.bmax[k], v[i][k]);
                    if (sh.normals) ts.n[i][k] = sh.normals[3 * vi + k];
                }
                if (sh.uvs) { ts.uv[i][0] = sh.uvs[2 * vi]; ts.uv[i][1] = sh.uvs[2 * vi + 1]; }
            }
            if (!sh.uvs) { // src/shapes/triangle_mesh.inl:86-90
                ts.uv[0][0] = 0; ts.uv[0][1] = 0; ts.uv[1][0] = 1; ts.uv[1][1] = 0; ts.uv[2][0] = 1; ts.uv[2][1] = 1;
            }
            ts.shape_id = s; ts.prim_id = t; ts.material_id = sh.material_id; ts.light_id = sh.area_light_id;
            ts.has_normals = sh.normals != nullptr; ts.has_uvs = sh.uvs != nullptr;
            for (int k = 0; k < 3; k++) { pr.v0[k] = v[0][k]; pr.e1[k] = v[1][k] - v[0][k]; pr.e2[k] = v[2][k] - v[0][k]; }
            for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) tri_verts.push_back(v[i][k]);
            gdpt::precompute_tri_constants(pos64, pr.e1, pr.e2, &ts);
            pr.gid = (uint32_t)tris.size();
            tris.push_back(ts); prim_in.push_back(pr); bounds.push_back(pb);
        }
    }
    if (tris.size() >= (size_t)GDPT_SPHERE_FLAG / 8) throw std::runtime_error("gdpt_scene_upload: too many triangles");
    for (int s = 0; s < desc->num_shapes; s++) {
        const GdptShape &sh = desc->shapes[s];
        if (sh.type != GDPT_SHAPE_SPHERE) continue;
        DevSphere sp{};
        DevPrim pr{};
        gdpt::PrimBounds pb;
        for (int k = 0; k < 3; k++) {
            sp.center[k] = sh.center[k];
            // scene bounds as Embree sees them: sphere_bounds_func stores double -> float (src/shapes/sphere.inl:1-10)
            lb[k] = std::min(lb[k], (float)(sh.center[k] - sh.radius)); ub[k] = std::max(ub[k], (float)(sh.center[k] + sh.radius));
            // BVH bounds: rounded outward so the box always contains the fp64 sphere
            pb.bmin[k] = std::nextafterf((float)(sh.center[k] - sh.radius), -std::numeric_limits<float>::infinity());
            pb.bmax[k] = std::nextafterf((float)(sh.center[k] + sh.radius), std::numeric_limits<float>::infinity());
        }
        sp.radius = sh.radius; sp.shape_id = s; sp.material_id = sh.material_id; sp.light_id = sh.area_light_id;
        pr.gid = GDPT_SPHERE_FLAG | (uint32_t)spheres.size();
        spheres.push_back(sp); prim_in.push_back(pr); bounds.push_back(pb);
    }

    // large triangles of big meshes are referenced from several smaller boxes (host/presplit.cpp); scenes small enough
    // for LDS keep one reference per primitive
    std::vector<gdpt::PrimBounds> refs;
    std::vector<uint32_t> ref_prim;
    {
        double budget = tris.size() >= 4096 ? kPresplitBudget : 0.0;
        budget = gdpt::debug_knob("presplit", budget);
        gdpt::presplit_triangles(bounds, tri_verts, budget, &refs, &ref_prim);
    }
    // spatial splits inside the SAH build (host/sbvh.cpp) for meshes that are walked from HBM; `sbvh` = extra references allowed
    // per primitive (test knob; 0 = the plain object-split build)
    const double sbvh_budget = gdpt::debug_knob("sbvh", tris.size() >= 4096 ? kSbvhBudget : 0.0);
    gdpt::BvhBuildResult bvh = sbvh_budget > 0 ? gdpt::build_sbvh(bounds, tri_verts, sbvh_budget, &ref_prim) : gdpt::build_bvh(refs);
    {   // widen every child box: the traversal's slab test then needs no per-test padding (device_trace.h: box_hit)
        float ext = 0.f;
        for (int k = 0; k < 3; k++) if (ub[k] >= lb[k]) ext = std::max(ext, std::max(std::fabs(ub[k]), std::fabs(lb[k])));
        for (auto &sp : spheres) for (int k = 0; k < 3; k++) ext = std::max(ext, (float)(std::fabs(sp.center[k]) + sp.radius));
        {   // ray origins: surface points (inside the bounds) and the camera position, xform_point(cam_to_world, 0)
            const double *m = cam.cam_to_world;
            for (int k = 0; k < 3; k++) ext = std::max(ext, (float)std::fabs(m[4 * k + 3] / m[15]) * 1.0000002f);
        }
        const float pad = ext * 1e-6f + 1e-30f;
        for (auto &n : bvh.nodes)
            for (int k = 0; k < 3; k++) {
                if (n.lmin[k] <= n.lmax[k]) { n.lmin[k] -= pad; n.lmax[k] += pad; }
                if (n.rmin[k] <= n.rmax[k]) { n.rmin[k] -= pad; n.rmax[k] += pad; }
            }
    }
    // wide form for scenes walked from HBM (same padded boxes); narrower nodes if the stack bound would not hold
    // (the 8-wide quantised form is built, verified and uploaded only by the GDPT_HBM_BVH8 A/B library: a product upload neither
    // pays for it nor can fail on it)
    gdpt::WideBvh wide = gdpt::collapse_for_traversal(bvh.nodes, GDPT_HBM_BVH8 != 0);
    if (wide.stack_need > GDPT_BVH_MAX_DEPTH) throw std::runtime_error("gdpt_scene_upload: BVH deeper than the traversal stack (builder bug)");
    if (wide.stack_need8 > GDPT_BVH_MAX_DEPTH + GDPT_STACK_OVERFLOW) throw std::runtime_error("gdpt_scene_upload: BVH8 deeper than the traversal stack (builder bug)");
    static_assert(sizeof(DevBvh8Node) == 128 && sizeof(DevBvh4Node) == 128, "wide BVH nodes are one 128-byte line");
    const std::vector<DevBvh4Node> &nodes4 = wide.nodes;
    sc->wide_stack_need = wide.stack_need;
    sc->wide8_stack_need = wide.stack_need8;
    std::vector<DevPrim> prims(bvh.order.size());
    for (size_t i = 0; i < bvh.order.size(); i++) prims[i] = prim_in[ref_prim[bvh.order[i]]];
    sc->bvh_depth = bvh.depth;
    if (bvh.depth > GDPT_BVH_MAX_DEPTH) throw std::runtime_error("gdpt_scene_upload: BVH deeper than the traversal stack (builder bug)");

    // ---- textures: fp64 mip chains exactly as make_mipmap builds them (src/mipmap.h:27-48) ----
    std::vector<DevImage> images;
    std::vector<double> texels;
    for (int i = 0; i < desc->num_images; i++) {
        const GdptImage &im = desc->images[i];
        if (im.width <= 0 || im.height <= 0 || (im.channels != 1 && im.channels != 3) || !im.texels)
            throw std::runtime_error("gdpt_scene_upload: bad image");
        DevImage di{};
        di.channels = im.channels;
        int size = std::max(im.width, im.height);
        int num_levels = std::min((int)std::ceil(std::log2((double)size) + 1), 8);
        di.num_levels = num_levels;
        int pw = im.width, ph = im.height;
        size_t prev_off = texels.size();
        di.width[0] = pw; di.height[0] = ph; di.offset[0] = (int64_t)prev_off;
        texels.insert(texels.end(), im.texels, im.texels + (size_t)pw * ph * im.channels);
        for (int l = 1; l < num_levels; l++) {
            int nw = std::max(pw / 2, 1), nh = std::max(ph / 2, 1);
            size_t off = texels.size();
            texels.resize(off + (size_t)nw * nh * im.channels);
            auto P = [&](int x, int y, int c) { x = std::min(x, pw - 1); y = std::min(y, ph - 1); return texels[prev_off + ((size_t)y * pw + x) * im.channels + c]; };
            for (int y = 0; y < nh; y++) for (int x = 0; x < nw; x++) for (int c = 0; c < im.channels; c++)
                texels[off + ((size_t)y * nw + x) * im.channels + c] =
                    (P(2 * x, 2 * y, c) + P(2 * x + 1, 2 * y, c) + P(2 * x, 2 * y + 1, c) + P(2 * x + 1, 2 * y + 1, c)) / 4.0;
            di.width[l] = nw; di.height[l] = nh; di.offset[l] = (int64_t)off;
            prev_off = off; pw = nw; ph = nh;
        }
        images.push_back(di);
    }

    std::vector<GdptMaterial> materials(desc->materials, desc->materials + desc->num_materials);
    std::vector<double> light_intensity;
    for (int l = 0; l < desc->num_lights; l++) for (int k = 0; k < 3; k++) light_intensity.push_back(desc->lights[l].intensity[k]);

    // ---- Integrator::Path emitter tables (same formulas and operation order as the reference) ----
    int env_power_slot = -1;
    std::vector<double> light_power;
    std::vector<DevLight> dlights;
    std::vector<double> light_pmf, light_cdf, light_tri_cdf, light_tri_pos, light_tri_nrm;
    auto table_1d = [](const std::vector<double> &f, std::vector<double> &pmf, std::vector<double> &cdf) {   // src/table_dist.cpp:3-25
        pmf = f;
        cdf.assign(f.size() + 1, 0.0);
        for (size_t i = 0; i < f.size(); i++) cdf[i + 1] = cdf[i] + pmf[i];
        const double total = cdf.back();
        if (total > 0) { for (size_t i = 0; i < pmf.size(); i++) { pmf[i] /= total; cdf[i] /= total; } }
        else {
            for (size_t i = 0; i < pmf.size(); i++) { pmf[i] = 1.0 / (double)pmf.size(); cdf[i] = (double)i / (double)pmf.size(); }
            cdf.back() = 1;
        }
    };
    {
        std::vector<int> sphere_index_of_shape((size_t)desc->num_shapes, -1);
        { int k = 0; for (int s = 0; s < desc->num_shapes; s++) if (desc->shapes[s].type == GDPT_SHAPE_SPHERE) sphere_index_of_shape[(size_t)s] = k++; }
        std::vector<double> power;
        for (int l = 0; l < desc->num_lights; l++) {
            const GdptLight &lt = desc->lights[l];
            if (lt.shape_id < 0 && desc->has_envmap && l == desc->envmap.light_id) {      // environment map: power filled in below
                dlights.push_back(DevLight{});
                power.push_back(0.0);
                continue;
            }
            if (lt.shape_id < 0 || lt.shape_id >= desc->num_shapes) throw std::runtime_error("gdpt_scene_upload: light without a shape");
            const GdptShape &sh = desc->shapes[lt.shape_id];
            DevLight dl{};
            for (int k = 0; k < 3; k++) dl.intensity[k] = lt.intensity[k];
            if (sh.type == GDPT_SHAPE_SPHERE) {
                dl.is_sphere = 1; dl.sphere_index = sphere_index_of_shape[(size_t)lt.shape_id];
                dl.area = 4 * 3.14159265358979323846 * sh.radius * sh.radius;                 // sphere.inl:207-209
            } else {
                dl.tri_first = (int)(light_tri_pos.size() / 9); dl.tri_count = sh.num_triangles;
                dl.cdf_first = (int)light_tri_cdf.size(); dl.has_normals = sh.normals ? 1 : 0;
                std::vector<double> areas((size_t)sh.num_triangles), pmf, cdf;
                double total = 0;
                for (int t = 0; t < sh.num_triangles; t++) {
                    const int *ix = sh.indices + 3 * t;
                    double p[3][3];
                    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { p[i][k] = sh.positions[3 * ix[i] + k]; light_tri_pos.push_back(p[i][k]); }
                    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) light_tri_nrm.push_back(sh.normals ? sh.normals[3 * ix[i] + k] : 0.0);
                    const double e1[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]}, e2[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
                    const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                    areas[(size_t)t] = std::sqrt(cx * cx + cy * cy + cz * cz) / 2;               // triangle_mesh.inl:70
                    total += areas[(size_t)t];
                }
                table_1d(areas, pmf, cdf);
                light_tri_cdf.insert(light_tri_cdf.end(), cdf.begin(), cdf.end());
                dl.area = total;
            }
            const double lum = lt.intensity[0] * 0.212671 + lt.intensity[1] * 0.715160 + lt.intensity[2] * 0.072169;   // src/spectrum.h:33-35
            power.push_back(lum * dl.area * 3.14159265358979323846);                              // diffuse_area_light.inl:1-3
            dlights.push_back(dl);
        }
        env_power_slot = desc->has_envmap ? desc->envmap.light_id : -1;
        light_power = power;
    }

    DevSceneView &v = sc->view;
    std::memcpy(v.cam.sample_to_cam, cam.sample_to_cam, sizeof(v.cam.sample_to_cam));
    std::memcpy(v.cam.cam_to_world, cam.cam_to_world, sizeof(v.cam.cam_to_world));
    {   // xform_point(cam_to_world, (0,0,0)), src/camera.cpp:42
        const double *m = cam.cam_to_world;
        double inv_w = 1.0 / m[15];
        v.cam.org[0] = m[3] * inv_w; v.cam.org[1] = m[7] * inv_w; v.cam.org[2] = m[11] * inv_w;
    }
    v.cam.width = cam.width; v.cam.height = cam.height; v.cam.filter_type = cam.filter_type; v.cam.filter_param = cam.filter_param;
    v.cam.pow2_film = ((cam.width & (cam.width - 1)) == 0 && (cam.height & (cam.height - 1)) == 0) ? 1 : 0;
    v.cam.inv_width = 1.0 / (double)cam.width; v.cam.inv_height = 1.0 / (double)cam.height;
    v.nodes = sc->keep(upload(bvh.nodes));
    v.nodes4 = sc->keep(upload(nodes4));
    v.nodes8 = sc->keep(upload(wide.nodes8));
    v.nodes4q = GDPT_HBM_Q4 ? sc->keep(upload(gdpt::quantise_bvh4(nodes4))) : nullptr;
    v.prims = sc->keep(upload(prims));
    v.tris = sc->keep(upload(tris));
    v.spheres = sc->keep(upload(spheres));
    v.materials = sc->keep(upload(materials));
    v.light_intensity = sc->keep(upload(light_intensity));
    v.images = sc->keep(upload(images));
    v.texels = sc->keep(upload(texels));
    v.lights = sc->keep(upload(dlights));
    v.light_tri_cdf = sc->keep(upload(light_tri_cdf));
    v.light_tri_pos = sc->keep(upload(light_tri_pos)); v.light_tri_nrm = sc->keep(upload(light_tri_nrm));
    sc->has_envmap = desc->has_envmap != 0;
    v.num_nodes = (int)bvh.nodes.size(); v.num_nodes4 = (int)nodes4.size(); v.num_nodes8 = (int)wide.nodes8.size(); v.num_prims = (int)prims.size();
    v.num_tris = (int)tris.size(); v.num_spheres = (int)spheres.size();
    v.num_materials = desc->num_materials; v.num_lights = desc->num_lights; v.num_images = desc->num_images;
    v.max_depth = desc->max_depth; v.rr_depth = desc->rr_depth;
    v.all_textures_constant = 1;
    for (auto &m : materials) for (auto &t : m.tex) if (t.type != GDPT_TEX_CONSTANT) v.all_textures_constant = 0;
    for (auto &m : materials) {
        sc->material_mask |= 1u << m.type;
        if (m.type != GDPT_MAT_LAMBERTIAN) sc->lambert_only = false;
        if (m.type == GDPT_MAT_ROUGHPLASTIC || m.type == GDPT_MAT_ROUGHDIELECTRIC) sc->has_rough = true;
        if (m.type == GDPT_MAT_DISNEY_GLASS || m.type == GDPT_MAT_DISNEY_BSDF || m.type == GDPT_MAT_ROUGHDIELECTRIC) sc->one_sided = false;   // two-sided lobes
    }
    // scenes with a refractive lobe (DisneyGlass, RoughDielectric): paths through glass are long-tailed, the work items are cut smaller
    // (render_kernels.hip: make_chunk_plan; disney_glass +9..15 %, matpreview's Integrator::Path +7 %; DisneyBSDF and the opaque scenes are flat)
    if (sc->material_mask & ((1u << GDPT_MAT_DISNEY_GLASS) | (1u << GDPT_MAT_ROUGHDIELECTRIC))) sc->plan_take_pct = 40;
    // get_intersection_epsilon (src/scene.h:100-102) from Embree-style fp32 scene bounds (src/scene.cpp:29-33)
    double dx = (double)ub[0] - (double)lb[0], dy = (double)ub[1] - (double)lb[1], dz = (double)ub[2] - (double)lb[2];
    double radius = prims.empty() ? 0.0 : std::sqrt(dx * dx + dy * dy + dz * dz) / 2;
    for (int k = 0; k < 3; k++) { sc->bounds[k] = lb[k]; sc->bounds[3 + k] = ub[k]; }
    v.isect_eps = std::min(radius * 1e-5, 0.01);

    // ---- environment map (Integrator::Path): TableDist2D over luminance * sin(elevation) of the level-0 image
    // (init_sampling_dist, src/lights/envmap.inl:66-83; make_table_dist_2d, src/table_dist.cpp:40-112), its power
    // (envmap.inl:1-5) and only then the light selection table (src/scene.cpp:44-53)
    v.has_envmap = 0; v.env_light_id = -1;
    if (desc->has_envmap) {
        const GdptEnvmap &e = desc->envmap;
        if (e.image_id < 0 || e.image_id >= desc->num_images || desc->images[e.image_id].channels != 3)
            throw std::runtime_error("gdpt_scene_upload: environment map without a 3-channel image");
        const GdptImage &im = desc->images[e.image_id];
        const int w = im.width, h = im.height;
        auto texel = [&](int x, int y) { const double *p = im.texels + ((size_t)y * w + x) * 3; return p; };
        auto modulo = [](int a, int b) { int r = a % b; return r < 0 ? r + b : r; };
        std::vector<double> f((size_t)w * h);
        size_t i = 0;
        for (int y = 0; y < h; y++) {
            const double vv = (y + 0.5) / (double)h;
            const double sin_elevation = std::sin(3.14159265358979323846 * vv);
            for (int x = 0; x < w; x++) {
                const double uu = (x + 0.5) / (double)w;
                // lookup(mipmap, u, v, 0): bilinear at level 0 with repeat wrap (src/mipmap.h:51-72)
                double u = uu * w - 0.5, vq = vv * h - 0.5;
                int ufi = modulo((int)u, w), vfi = modulo((int)vq, h);
                int uci = modulo(ufi + 1, w), vci = modulo(vfi + 1, h);
                double u_off = u - ufi, v_off = vq - vfi;
                double rgb[3];
                for (int c = 0; c < 3; c++)
                    rgb[c] = texel(ufi, vfi)[c] * (1 - u_off) * (1 - v_off) + texel(ufi, vci)[c] * (1 - u_off) * v_off +
                             texel(uci, vfi)[c] * u_off * (1 - v_off) + texel(uci, vci)[c] * u_off * v_off;
                f[i++] = (rgb[0] * 0.212671 + rgb[1] * 0.715160 + rgb[2] * 0.072169) * sin_elevation;
            }
        }
        std::vector<double> cdf_rows((size_t)h * (w + 1)), pdf_rows((size_t)h * w), cdf_m((size_t)h + 1), pdf_m((size_t)h);
        for (int y = 0; y < h; y++) {
            double *cdf = &cdf_rows[(size_t)y * (w + 1)];
            cdf[0] = 0;
            for (int x = 0; x < w; x++) cdf[x + 1] = cdf[x] + f[(size_t)y * w + x];
            const double integral = cdf[w];
            if (integral > 0) {
                for (int x = 0; x < w; x++) cdf[x] /= integral;
                for (int x = 0; x < w; x++) pdf_rows[(size_t)y * w + x] = f[(size_t)y * w + x] / integral;
            } else {
                for (int x = 0; x < w; x++) { pdf_rows[(size_t)y * w + x] = 1.0 / (double)w; cdf[x] = (double)x / (double)w; }
                cdf[w] = 1;
            }
        }
        cdf_m[0] = 0;
        for (int y = 0; y < h; y++) cdf_m[(size_t)y + 1] = cdf_m[(size_t)y] + cdf_rows[(size_t)y * (w + 1) + w];
        const double total_values = cdf_m.back();
        if (total_values > 0) {
            for (int y = 0; y < h; y++) cdf_m[(size_t)y] /= total_values;
            cdf_m[(size_t)h] = 1;
            for (int y = 0; y < h; y++) pdf_m[(size_t)y] = cdf_rows[(size_t)y * (w + 1) + w] / total_values;
        } else {
            for (int y = 0; y < h; y++) { pdf_m[(size_t)y] = 1.0 / (double)h; cdf_m[(size_t)y] = (double)y / (double)h; }
            cdf_m[(size_t)h] = 1;
        }
        for (int y = 0; y < h; y++) cdf_rows[(size_t)y * (w + 1) + w] = 1;
        v.has_envmap = 1; v.env_light_id = e.light_id; v.env_image_id = e.image_id; v.env_w = w; v.env_h = h; v.env_scale = e.scale;
        std::memcpy(v.env_to_world, e.to_world, sizeof(v.env_to_world));
        std::memcpy(v.env_to_local, e.to_local, sizeof(v.env_to_local));
        v.env_cdf_rows = sc->keep(upload(cdf_rows)); v.env_pdf_rows = sc->keep(upload(pdf_rows));
        v.env_cdf_marginals = sc->keep(upload(cdf_m)); v.env_pdf_marginals = sc->keep(upload(pdf_m));
        if (env_power_slot >= 0 && env_power_slot < (int)light_power.size())
            light_power[(size_t)env_power_slot] = 3.14159265358979323846 * radius * radius * total_values / ((double)w * (double)h);
    }
    if (!light_power.empty()) table_1d(light_power, light_pmf, light_cdf);
    v.light_pmf = sc->keep(upload(light_pmf)); v.light_cdf = sc->keep(upload(light_cdf));

    ck(hipMalloc((void **)&sc->d_counters, sizeof(gdpt::RenderCounters)), "hipMalloc(counters)");
    ck(hipMalloc((void **)&sc->d_queue, sizeof(unsigned long long)), "hipMalloc(queue)");
    { hipDeviceProp_t prop; ck(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties"); sc->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
    ck(hipHostMalloc((void **)&sc->h_counters, sizeof(gdpt::RenderCounters)), "hipHostMalloc(counters)");
    ck(hipEventCreate(&sc->ev0), "hipEventCreate");
    ck(hipEventCreate(&sc->ev1), "hipEventCreate");
}

} // namespace gdpt

namespace {

using gdpt::build_scene;

struct Band { int spp, rng, row_begin, row_end, max_depth, shift, plan_rows; };

Band resolve(const GdptScene *sc, const GdptRenderParams *p) {
    if (!sc) throw std::runtime_error("null scene handle");
    Band b;
    b.spp = (p && p->spp > 0) ? p->spp : 0;
    b.rng = p ? p->rng_scheme : GDPT_RNG_SAMPLE;
    b.row_begin = p ? p->row_begin : 0;
    b.row_end = p ? p->row_end : 0;
    if (b.row_begin == 0 && b.row_end == 0) b.row_end = sc->view.cam.height;
    if (b.row_begin < 0 || b.row_end > sc->view.cam.height || b.row_begin >= b.row_end) throw std::runtime_error("gdpt_render: bad row band");
    if (b.rng != GDPT_RNG_TILE && b.rng != GDPT_RNG_SAMPLE) throw std::runtime_error("gdpt_render: unknown rng_scheme");
    b.shift = p ? p->shift_mode : GDPT_SHIFT_REFERENCE;
    if (b.shift != GDPT_SHIFT_REFERENCE && b.shift != GDPT_SHIFT_RECONNECT) throw std::runtime_error("gdpt_render: unknown shift_mode");
    if (b.shift == GDPT_SHIFT_RECONNECT && b.rng != GDPT_RNG_SAMPLE) throw std::runtime_error("gdpt_render: GDPT_SHIFT_RECONNECT needs GDPT_RNG_SAMPLE");
    if (b.rng == GDPT_RNG_TILE && ((b.row_begin % 16) != 0 || ((b.row_end % 16) != 0 && b.row_end != sc->view.cam.height)))
        throw std::runtime_error("gdpt_render: GDPT_RNG_TILE bands must cover whole 16-pixel tile rows");
    b.max_depth = (p && p->max_depth_override != 0) ? p->max_depth_override : sc->view.max_depth;
    b.plan_rows = (p && p->plan_rows > 0) ? std::min(p->plan_rows, sc->view.cam.height) : sc->view.cam.height;
    if (p && p->plan_rows < 0) throw std::runtime_error("gdpt_render: plan_rows must be >= 0");
    return b;
}

} // namespace

namespace gdpt {
// Enqueues one render; returns after enqueue unless stats are requested.
void render_device_impl(GdptScene *sc, const GdptRenderParams *params, int scene_spp,
                        double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                        hipStream_t stream, GdptRenderStats *stats) {
    ck(hipSetDevice(sc->device), "hipSetDevice");
    Band b = resolve(sc, params);
    if (b.spp <= 0) b.spp = scene_spp;
    if (b.spp <= 0) throw std::runtime_error("gdpt_render: samples per pixel must be > 0");
    if (!img || !cx0 || !cy0 || !cx1 || !cy1) throw std::runtime_error("gdpt_render: null output buffer");
    gdpt::RenderLaunch rl{};
    rl.spp = b.spp; rl.rng_scheme = b.rng; rl.row_begin = b.row_begin; rl.row_end = b.row_end; rl.max_depth = b.max_depth; rl.shift_mode = b.shift; rl.plan_rows = b.plan_rows;
    rl.img = img; rl.cx0 = cx0; rl.cy0 = cy0; rl.cx1 = cx1; rl.cy1 = cy1;
    rl.counters = sc->d_counters;
    rl.count_traversal = stats && stats->nodes_visited == ~0ull;   // request flag: caller presets nodes_visited = UINT64_MAX
    rl.one_sided_materials = sc->one_sided && !sc->has_rough; rl.lambert_only = sc->lambert_only;
    rl.material_mask = sc->material_mask;
    rl.wide_stack_need = GDPT_HBM_BVH8 ? std::min(sc->wide8_stack_need, GDPT_BVH_MAX_DEPTH) : sc->wide_stack_need;   // LDS slots; the BVH8 may go on in private memory
    rl.num_materials = sc->view.num_materials;
    rl.scene_fits_lds = gdpt::scene_fits_lds(sc->view.num_nodes, sc->view.num_prims, sc->view.num_tris, sc->view.num_materials, sc->view.num_lights, sc->bvh_depth);
    // A/B overrides of the parity tests (include/gdpt_debug.h); every default below is the product path
    auto env_int = [](const char *name, int def) { return gdpt::debug_knob_int(name, def); };
    if (env_int("full_material_switch", 0)) rl.material_mask = 0x1FFu;
    rl.no_spheres = sc->view.num_spheres == 0 && !env_int("no_plain_kernel", 0);
    rl.const_textures = sc->view.all_textures_constant != 0;
    rl.replay_per_step = env_int("replay_per_step", 0);
    rl.force_eager = env_int("force_eager", 0) != 0;
    rl.thresh_a = env_int("keep_frac", -1); rl.thresh_c = env_int("search_frac", -1);
    rl.force_log2k = env_int("log2k", -1);
    rl.num_cus = sc->num_cus;
    rl.plan_take_pct = sc->plan_take_pct;
    rl.blocks_per_cu = env_int("blocks_per_cu", 0);
    rl.stamped = env_int("stamps", 0) != 0;
    {
        size_t need = gdpt::render_partials_doubles(sc->view.cam.width, b.row_end - b.row_begin, b.plan_rows, b.spp, rl.force_log2k, (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2) * 256, rl.plan_take_pct);
        if (need > sc->partials_doubles) {
            if (sc->d_partials) { ck(hipStreamSynchronize(stream), "hipStreamSynchronize"); hipFree(sc->d_partials); sc->d_partials = nullptr; }
            ck(hipMalloc((void **)&sc->d_partials, need * sizeof(double)), "hipMalloc(work-item partials)");
            sc->partials_doubles = need;
        }
        rl.partials = sc->d_partials; rl.queue_head = sc->d_queue;
    }
    if (env_int("no_lds_scene", 0)) rl.scene_fits_lds = false;
    // two-sided lobes (DisneyGlass, DisneyBSDF) without rough ones: lane machine with offsets replayed from a bounce log
    rl.two_sided_machine = !sc->one_sided && !sc->has_rough && !rl.force_eager && b.rng == GDPT_RNG_SAMPLE && !env_int("no_twosided_machine", 0);
    if (rl.two_sided_machine) {
        const long long tiles = (long long)((sc->view.cam.width + 15) / 16) * ((b.row_end - b.row_begin + 15) / 16);
        const long long items = (tiles * 256) * gdpt::make_chunk_plan(b.spp, rl.force_log2k, (long long)sc->view.cam.width * b.plan_rows,
                                                                      (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2) * 256, rl.plan_take_pct).n;
        const size_t need = gdpt::twosided_log_bytes(gdpt::persistent_blocks(rl, items));
        if (need > sc->bounce_log_bytes) {
            if (sc->d_bounce_log) { ck(hipStreamSynchronize(stream), "hipStreamSynchronize"); hipFree(sc->d_bounce_log); sc->d_bounce_log = nullptr; }
            ck(hipMalloc(&sc->d_bounce_log, need), "hipMalloc(bounce log)");
            sc->bounce_log_bytes = need;
        }
        rl.bounce_log = sc->d_bounce_log; rl.bounce_log_bytes = sc->bounce_log_bytes;
    }
    // scenes walked from HBM with one-sided lobes: the wavefront pipeline (test knob "wavefront": 0 = lane machine, 1 = wavefront)
    rl.wavefront = env_int("wavefront", kWavefrontDefault) != 0 && rl.one_sided_materials && !rl.scene_fits_lds && !rl.force_eager &&
                   b.rng == GDPT_RNG_SAMPLE && b.shift == GDPT_SHIFT_REFERENCE;
    if (rl.wavefront) {
        const long long tiles = (long long)((sc->view.cam.width + 15) / 16) * ((b.row_end - b.row_begin + 15) / 16);
        const long long items = (tiles * 256) * gdpt::make_chunk_plan(b.spp, rl.force_log2k, (long long)sc->view.cam.width * b.plan_rows,
                                                                      (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2) * 256, rl.plan_take_pct).n;
        int slots = gdpt::wf_slot_count(items);
        { const int forced = env_int("wf_slots", 0); if (forced > 0) slots = std::min(slots, (forced + 255) / 256 * 256); }   // tests: force slot reuse
        if (slots > sc->wf_slots) {
            ck(hipStreamSynchronize(stream), "hipStreamSynchronize");
            if (sc->d_wf_state) hipFree(sc->d_wf_state);
            if (sc->d_wf_live) hipFree(sc->d_wf_live);
            if (sc->d_wf_aux) hipFree(sc->d_wf_aux);
            sc->d_wf_state = nullptr; sc->d_wf_live = nullptr; sc->d_wf_aux = nullptr; sc->wf_slots = 0;
            ck(hipMalloc((void **)&sc->d_wf_state, (size_t)slots * gdpt::wf_words() * sizeof(unsigned long long)), "hipMalloc(wavefront state)");
            ck(hipMalloc((void **)&sc->d_wf_live, (size_t)slots * sizeof(unsigned)), "hipMalloc(wavefront live list)");
            ck(hipMalloc(&sc->d_wf_aux, gdpt::wf_aux_bytes(slots)), "hipMalloc(wavefront ray / hit records)");
            sc->wf_slots = slots;
        }
        if (!sc->d_wf_counters) {
            ck(hipMalloc((void **)&sc->d_wf_counters, sizeof(unsigned) * 3 * gdpt::wf_max_generations()), "hipMalloc(wavefront counters)");
            ck(hipHostMalloc((void **)&sc->h_wf_word, sizeof(unsigned)), "hipHostMalloc(wavefront)");
            ck(hipEventCreateWithFlags(&sc->wf_event, hipEventDisableTiming), "hipEventCreate");
        }
        rl.wf_aux = sc->d_wf_aux; rl.wf_sort = env_int("wf_sort", 1);
        for (int k = 0; k < 6; k++) rl.wf_bounds[k] = sc->bounds[k];
        rl.wf_state = sc->d_wf_state; rl.wf_live = sc->d_wf_live; rl.wf_counters = sc->d_wf_counters; rl.wf_host = sc->h_wf_word;
        rl.wf_event = sc->wf_event; rl.wf_slots = slots;       // exactly the slots this band needs (the buffers may be larger)
    }
    rl.lds_wide = rl.scene_fits_lds && env_int("lds_wide", 1) != 0 &&
                  gdpt::scene_fits_lds_wide(sc->view.num_nodes4, sc->view.num_prims, sc->view.num_tris, sc->view.num_materials, sc->view.num_lights, sc->wide_stack_need);
    ck(hipMemsetAsync(sc->d_counters, 0, sizeof(gdpt::RenderCounters), stream), "hipMemsetAsync(counters)");
    if (rl.stamped) ck(hipMemsetAsync(&sc->d_counters->stamps[12], 0xFF, 2 * sizeof(unsigned long long), stream), "hipMemsetAsync(stamps)");   // min slots
    if (stats) ck(hipEventRecord(sc->ev0, stream), "hipEventRecord");
    gdpt::launch_render(sc->view, rl, stream);
    if (stats) {
        ck(hipEventRecord(sc->ev1, stream), "hipEventRecord");
        ck(hipMemcpyAsync(sc->h_counters, sc->d_counters, sizeof(gdpt::RenderCounters), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(counters)");
        ck(hipStreamSynchronize(stream), "hipStreamSynchronize(render)");
        float ms = 0;
        ck(hipEventElapsedTime(&ms, sc->ev0, sc->ev1), "hipEventElapsedTime");
        stats->samples = (uint64_t)sc->view.cam.width * (uint64_t)(b.row_end - b.row_begin) * (uint64_t)b.spp;
        stats->rays = sc->h_counters->rays; stats->bounces = sc->h_counters->bounces;
        stats->nonfinite_samples = sc->h_counters->nonfinite;
        stats->nodes_visited = sc->h_counters->nodes; stats->tris_tested = sc->h_counters->prims;
        stats->render_ms = ms;
        stats->wave_node_trips = sc->h_counters->wave_node_trips; stats->wave_leaf_trips = sc->h_counters->wave_leaf_trips;
        stats->wave_steps = sc->h_counters->wave_steps; stats->lane_steps = sc->h_counters->lane_steps;
        if (rl.stamped) gdpt::debug_store_stamps(sc->h_counters->stamps, 16);
        // only the persistent kernel over an LDS-resident scene can still walk the BVH2 form
        const bool lds_kernel = rl.one_sided_materials && !rl.force_eager && b.rng == GDPT_RNG_SAMPLE && rl.scene_fits_lds;
        stats->node_bytes = (lds_kernel && !rl.lds_wide) ? sizeof(DevBvhNode) : sizeof(DevBvh4Node);
    }
}

} // namespace gdpt

namespace {
using gdpt::render_device_impl;
// Integrator::Path: enqueues one render of `img`; returns after enqueue unless stats are requested.
void path_render_device_impl(GdptScene *sc, const GdptRenderParams *params, double *img, hipStream_t stream, GdptRenderStats *stats) {
    ck(hipSetDevice(sc->device), "hipSetDevice");
    if (sc->view.num_lights <= 0) throw std::runtime_error("gdpt_path_render: the scene has no emitter to sample");
    Band b = resolve(sc, params);
    if (b.spp <= 0) b.spp = sc->scene_spp;
    if (b.spp <= 0) throw std::runtime_error("gdpt_path_render: samples per pixel must be > 0");
    if (!img) throw std::runtime_error("gdpt_path_render: null output buffer");
    gdpt::RenderLaunch rl{};
    rl.spp = b.spp; rl.rng_scheme = b.rng; rl.row_begin = b.row_begin; rl.row_end = b.row_end; rl.max_depth = b.max_depth; rl.shift_mode = b.shift; rl.plan_rows = b.plan_rows;
    rl.img = img;
    rl.counters = sc->d_counters;
    rl.count_traversal = stats && stats->nodes_visited == ~0ull;
    auto env_int = [](const char *name, int def) { return gdpt::debug_knob_int(name, def); };   // include/gdpt_debug.h
    rl.force_log2k = env_int("log2k", -1);
    rl.force_eager = env_int("force_eager", 0) != 0;
    rl.thresh_a = env_int("keep_frac", -1); rl.thresh_c = env_int("search_frac", -1);
    rl.num_cus = sc->num_cus; rl.blocks_per_cu = env_int("blocks_per_cu", 0);
    rl.lambert_only = sc->lambert_only;
    rl.no_spheres = sc->view.num_spheres == 0 && !env_int("no_plain_kernel", 0); rl.const_textures = sc->view.all_textures_constant != 0;
    rl.scene_fits_lds = !env_int("no_lds_scene", 0) &&
                        gdpt::scene_fits_lds_wide(sc->view.num_nodes4, sc->view.num_prims, sc->view.num_tris, sc->view.num_materials, sc->view.num_lights, sc->wide_stack_need);
    {
        size_t need = gdpt::render_partials_doubles(sc->view.cam.width, b.row_end - b.row_begin, b.plan_rows, b.spp, rl.force_log2k, (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2) * 256, rl.plan_take_pct);
        if (need > sc->partials_doubles) {
            if (sc->d_partials) { ck(hipStreamSynchronize(stream), "hipStreamSynchronize"); hipFree(sc->d_partials); sc->d_partials = nullptr; }
            ck(hipMalloc((void **)&sc->d_partials, need * sizeof(double)), "hipMalloc(work-item partials)");
            sc->partials_doubles = need;
        }
        rl.partials = sc->d_partials; rl.queue_head = sc->d_queue;
    }
    ck(hipMemsetAsync(sc->d_counters, 0, sizeof(gdpt::RenderCounters), stream), "hipMemsetAsync(counters)");
    if (stats) ck(hipEventRecord(sc->ev0, stream), "hipEventRecord");
    gdpt::launch_path_render(sc->view, rl, stream);
    if (stats) {
        ck(hipEventRecord(sc->ev1, stream), "hipEventRecord");
        ck(hipMemcpyAsync(sc->h_counters, sc->d_counters, sizeof(gdpt::RenderCounters), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(counters)");
        ck(hipStreamSynchronize(stream), "hipStreamSynchronize(path render)");
        float ms = 0;
        ck(hipEventElapsedTime(&ms, sc->ev0, sc->ev1), "hipEventElapsedTime");
        std::memset(stats, 0, sizeof(*stats));
        stats->samples = (uint64_t)sc->view.cam.width * (uint64_t)(b.row_end - b.row_begin) * (uint64_t)b.spp;
        stats->rays = sc->h_counters->rays; stats->bounces = sc->h_counters->bounces;
        stats->nonfinite_samples = sc->h_counters->nonfinite;
        stats->nodes_visited = sc->h_counters->nodes; stats->tris_tested = sc->h_counters->prims;
        stats->render_ms = ms;
        stats->node_bytes = sizeof(DevBvh4Node);
    }
}

} // namespace

extern "C" {

#ifndef GDPT_BUILD_ARCH
#error "GDPT_BUILD_ARCH must name the --offload-arch the kernels were compiled for (csrc/Makefile passes it)"
#endif
const char *gdpt_build_arch(void) { return GDPT_BUILD_ARCH; }

// include/gdpt_debug.h: the work-item plan of the persistent kernels, for host-side tests (no GPU needed)
int gdpt_debug_chunk_plan(int spp, int force_log2k, long long film_pixels, long long resident_lanes, int32_t *begin, int capacity) {
    const gdpt::ChunkPlan p = gdpt::make_chunk_plan(spp, force_log2k, film_pixels, resident_lanes);
    if (!begin || capacity < p.n + 1) return -1;
    for (int c = 0; c <= p.n; c++) begin[c] = p.begin[c];
    return p.n;
}

int gdpt_scene_upload(const GdptSceneDesc *desc, int device, GdptScene **out_scene) {
    return gdpt::guarded([&]() {
        if (!out_scene) throw std::runtime_error("gdpt_scene_upload: null output");
        std::unique_ptr<GdptScene> sc(new GdptScene());
        build_scene(desc, device, sc.get());
        sc->scene_spp = desc->samples_per_pixel;
        *out_scene = sc.release();
    });
}

void gdpt_scene_free(GdptScene *scene) {
    if (!scene) return;
    delete scene;
}

int gdpt_scene_info(const GdptScene *scene, int32_t *num_nodes, int32_t *num_tris, int32_t *num_spheres, int32_t *bvh_depth) {
    return gdpt::guarded([&]() {
        if (!scene) throw std::runtime_error("null scene handle");
        if (num_nodes) *num_nodes = scene->view.num_nodes;
        if (num_tris) *num_tris = scene->view.num_tris;
        if (num_spheres) *num_spheres = scene->view.num_spheres;
        if (bvh_depth) *bvh_depth = scene->bvh_depth;
    });
}

int gdpt_render_device(GdptScene *scene, const GdptRenderParams *params,
                       double *d_img, double *d_cx0, double *d_cy0, double *d_cx1, double *d_cy1,
                       void *stream, GdptRenderStats *stats) {
    return gdpt::guarded([&]() {
        if (!scene) throw std::runtime_error("null scene handle");
        render_device_impl(scene, params, scene->scene_spp, d_img, d_cx0, d_cy0, d_cx1, d_cy1, (hipStream_t)stream, stats);
    });
}

int gdpt_render(GdptScene *scene, const GdptRenderParams *params,
                double *img, double *cx0, double *cy0, double *cx1, double *cy1, GdptRenderStats *stats) {
    return gdpt::guarded([&]() {
        if (!scene) throw std::runtime_error("null scene handle");
        if (!img || !cx0 || !cy0 || !cx1 || !cy1) throw std::runtime_error("gdpt_render: null output buffer");
        ck(hipSetDevice(scene->device), "hipSetDevice");
        size_t elems = (size_t)scene->view.cam.width * scene->view.cam.height * 3;
        scene->ensure_buffers(elems);
        Band b = resolve(scene, params);
        double *host[5] = {img, cx0, cy0, cx1, cy1};
        // rows outside the band keep the caller's contents: upload them first when rendering a partial band
        bool partial = (b.row_begin != 0 || b.row_end != scene->view.cam.height);
        if (partial) for (int k = 0; k < 5; k++) ck(hipMemcpy(scene->d_buf[k], host[k], elems * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(H2D)");
        GdptRenderStats local{};
        render_device_impl(scene, params, scene->scene_spp, scene->d_buf[0], scene->d_buf[1], scene->d_buf[2], scene->d_buf[3], scene->d_buf[4],
                           nullptr, stats ? stats : &local);
        for (int k = 0; k < 5; k++) ck(hipMemcpy(host[k], scene->d_buf[k], elems * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    });
}

int gdpt_tile_row_costs(GdptScene *scene, int spp, double *cost, int capacity) {
    return gdpt::guarded([&]() {
        if (!scene || !cost) throw std::runtime_error("gdpt_tile_row_costs: null argument");
        const int H = scene->view.cam.height, T = (H + 15) / 16;
        if (capacity < T) throw std::runtime_error("gdpt_tile_row_costs: capacity < ceil(height / 16)");
        ck(hipSetDevice(scene->device), "hipSetDevice");
        scene->ensure_buffers((size_t)scene->view.cam.width * H * 3);
        for (int t = 0; t < T; t++) {       // one small launch per tile row; the counters are exact, so is the cost
            GdptRenderParams p{};
            p.spp = spp > 0 ? spp : 1; p.rng_scheme = GDPT_RNG_SAMPLE; p.row_begin = t * 16; p.row_end = std::min(H, t * 16 + 16);
            GdptRenderStats st{};
            render_device_impl(scene, &p, scene->scene_spp, scene->d_buf[0], scene->d_buf[1], scene->d_buf[2], scene->d_buf[3], scene->d_buf[4], nullptr, &st);
            cost[t] = (double)st.rays;
        }
    });
}

int gdpt_path_render_device(GdptScene *scene, const GdptRenderParams *params, double *d_img, void *stream, GdptRenderStats *stats) {
    return gdpt::guarded([&]() {
        if (!scene) throw std::runtime_error("null scene handle");
        path_render_device_impl(scene, params, d_img, (hipStream_t)stream, stats);
    });
}

int gdpt_path_render(GdptScene *scene, const GdptRenderParams *params, double *img, GdptRenderStats *stats) {
    return gdpt::guarded([&]() {
        if (!scene) throw std::runtime_error("null scene handle");
        if (!img) throw std::runtime_error("gdpt_path_render: null output buffer");
        ck(hipSetDevice(scene->device), "hipSetDevice");
        size_t elems = (size_t)scene->view.cam.width * scene->view.cam.height * 3;
        scene->ensure_buffers(elems);
        Band b = resolve(scene, params);
        const bool partial = (b.row_begin != 0 || b.row_end != scene->view.cam.height);
        if (partial) ck(hipMemcpy(scene->d_buf[0], img, elems * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(H2D)");
        GdptRenderStats local{};
        path_render_device_impl(scene, params, scene->d_buf[0], nullptr, stats ? stats : &local);
        ck(hipMemcpy(img, scene->d_buf[0], elems * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    });
}

int gdpt_assemble_rows_device(int width, int height, int row_begin, int row_end, const double *d_img, const double *d_cx0, const double *d_cy0,
                              const double *d_cx1, const double *d_cy1, double *d_c, double *d_cx, double *d_cy, void *stream) {
    return gdpt::guarded([&]() {
        if (!d_img || !d_cx0 || !d_cy0 || !d_cx1 || !d_cy1 || !d_c || !d_cx || !d_cy) throw std::runtime_error("gdpt_assemble_device: null buffer");
        gdpt::launch_assemble(width, height, row_begin, row_end, d_img, d_cx0, d_cy0, d_cx1, d_cy1, d_c, d_cx, d_cy, (hipStream_t)stream);
    });
}
int gdpt_assemble_device(int width, int height, const double *d_img, const double *d_cx0, const double *d_cy0,
                         const double *d_cx1, const double *d_cy1, double *d_c, double *d_cx, double *d_cy, void *stream) {
    return gdpt_assemble_rows_device(width, height, 0, 0, d_img, d_cx0, d_cy0, d_cx1, d_cy1, d_c, d_cx, d_cy, stream);
}

int gdpt_assemble_solve_device(int width, int height, const double *d_img, const double *d_cx0, const double *d_cy0, const double *d_cx1, const double *d_cy1,
                               double *d_c, double *d_cx, double *d_cy, double dataCost, double *d_out, int solver, double tol, int max_iters,
                               void *stream, GdptPoissonStats *stats) {
    return gdpt::guarded([&]() {
        if (!d_img || !d_cx0 || !d_cy0 || !d_cx1 || !d_cy1 || !d_c || !d_cx || !d_cy || !d_out) throw std::runtime_error("gdpt_assemble_solve_device: null buffer");
        gdpt::PoissonResult r = gdpt::assemble_solve_device(width, height, d_img, d_cx0, d_cy0, d_cx1, d_cy1, d_c, d_cx, d_cy, dataCost, d_out, solver, tol, max_iters,
                                                            (hipStream_t)stream, stats != nullptr);
        if (stats) { stats->iterations = r.iterations; stats->solver = r.solver; stats->rel_residual = r.rel_residual; stats->solve_ms = r.solve_ms; }
    });
}

int gdpt_poisson_forget_stream(void *stream) {
    return gdpt::guarded([&]() {
        int dev = 0;
        ck(hipGetDevice(&dev), "hipGetDevice");
        gdpt::poisson_forget_stream(dev, (hipStream_t)stream);
    });
}

int gdpt_poisson_solve_device(int width, int height, const double *d_c, const double *d_gx, const double *d_gy,
                              double dataCost, double *d_out, int solver, double tol, int max_iters,
                              void *stream, GdptPoissonStats *stats) {
    return gdpt::guarded([&]() {
        if (!d_c || !d_gx || !d_gy || !d_out) throw std::runtime_error("gdpt_poisson_solve_device: null buffer");
        gdpt::PoissonResult r = gdpt::poisson_solve_device(width, height, d_c, d_gx, d_gy, dataCost, d_out, solver, tol, max_iters, (hipStream_t)stream, stats != nullptr);
        if (stats) { stats->iterations = r.iterations; stats->solver = r.solver; stats->rel_residual = r.rel_residual; stats->solve_ms = r.solve_ms; }
    });
}

int gdpt_poisson_solve_ex(int width, int height, const double *imgData, const double *imgGradX, const double *imgGradY,
                          double dataCost, double *imgOut, int solver, double tol, int max_iters, GdptPoissonStats *stats) {
    return gdpt::guarded([&]() {
        if (!imgData || !imgGradX || !imgGradY || !imgOut) throw std::runtime_error("gdpt_poisson_solve: null buffer");
        if (width <= 0 || height <= 0) throw std::runtime_error("gdpt_poisson_solve: empty image");
        int ndev = 0;
        ck(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (ndev <= 0) throw std::runtime_error("gdpt_poisson_solve: no HIP device visible (this library has no CPU fallback)");
        size_t bytes = (size_t)width * height * 3 * sizeof(double);
        double *d[4] = {nullptr, nullptr, nullptr, nullptr};
        auto cleanup = [&]() { for (auto p : d) if (p) hipFree(p); };
        try {
            for (auto &p : d) ck(hipMalloc((void **)&p, bytes), "hipMalloc(poisson io)");
            ck(hipMemcpy(d[0], imgData, bytes, hipMemcpyHostToDevice), "hipMemcpy");
            ck(hipMemcpy(d[1], imgGradX, bytes, hipMemcpyHostToDevice), "hipMemcpy");
            ck(hipMemcpy(d[2], imgGradY, bytes, hipMemcpyHostToDevice), "hipMemcpy");
            gdpt::PoissonResult r = gdpt::poisson_solve_device(width, height, d[0], d[1], d[2], dataCost, d[3], solver, tol, max_iters, nullptr, stats != nullptr);
            ck(hipMemcpy(imgOut, d[3], bytes, hipMemcpyDeviceToHost), "hipMemcpy");
            if (stats) { stats->iterations = r.iterations; stats->solver = r.solver; stats->rel_residual = r.rel_residual; stats->solve_ms = r.solve_ms; }
        } catch (...) { cleanup(); throw; }
        cleanup();
    });
}

int gdpt_poisson_solve(int width, int height, const double *imgData, const double *imgGradX, const double *imgGradY,
                       double dataCost, double *imgOut) {
    return gdpt_poisson_solve_ex(width, height, imgData, imgGradX, imgGradY, dataCost, imgOut, GDPT_SOLVER_DEFAULT, 0.0, 0, nullptr);
}

int gdpt_gradient_path_render(GdptScene *scene, const GdptRenderParams *params, double dataCost, double *out_image,
                              double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                              GdptRenderStats *rstats, GdptPoissonStats *pstats) {
    return gdpt::guarded([&]() {
        if (!scene || !out_image) throw std::runtime_error("gdpt_gradient_path_render: null argument");
        ck(hipSetDevice(scene->device), "hipSetDevice");
        const int w = scene->view.cam.width, h = scene->view.cam.height;
        size_t elems = (size_t)w * h * 3;
        scene->ensure_buffers(elems);
        double **b = scene->d_buf;
        GdptRenderParams p = params ? *params : GdptRenderParams{};
        p.row_begin = 0; p.row_end = 0;   // the solve is global: whole image only
        GdptRenderStats local{};
        render_device_impl(scene, &p, scene->scene_spp, b[0], b[1], b[2], b[3], b[4], nullptr, rstats ? rstats : &local);
        gdpt::PoissonResult r = gdpt::assemble_solve_device(w, h, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], dataCost, b[8], GDPT_SOLVER_DEFAULT, 0.0, 0, nullptr, pstats != nullptr);
        if (pstats) { pstats->iterations = r.iterations; pstats->solver = r.solver; pstats->rel_residual = r.rel_residual; pstats->solve_ms = r.solve_ms; }
        ck(hipMemcpy(out_image, b[8], elems * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
        double *host[5] = {img, cx0, cy0, cx1, cy1};
        for (int k = 0; k < 5; k++) if (host[k]) ck(hipMemcpy(host[k], b[k], elems * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    });
}

} // extern "C"
