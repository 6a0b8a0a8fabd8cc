// ref_kat.cpp — known-answer generator driven by the REFERENCE's own code.
//
// Built only in the development container (oracle/Makefile target `ref`): it #includes the reference headers
// and links the reference translation units where they lie under /root/reference (material.cpp, camera.cpp,
// filter.cpp, transform.cpp, shape.cpp, table_dist.cpp, parsers/parse_obj.cpp). Nothing of the reference is
// copied into this repository and nothing stands in for Embree: the functions that need it (intersect,
// register_embree, Scene::Scene) are never referenced from here and are dropped by --gc-sections.
//
// Output: one JSON document on stdout with inputs AND reference outputs, committed as
// tests/golden/ref_kat.json by oracle/gen_golden.py. Everything here is this repository's own driver code.
#include "camera.h"
#include "filter.h"
#include "intersection.h"
#include "material.h"
#include "parsers/parse_obj.h"
#include "parsers/shape_utils.h"
#include "pcg.h"
#include "point_and_normal.h"
#include "shape.h"
#include "spectrum.h"
#include "table_dist.h"
#include "texture.h"
#include "transform.h"

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

// defined in the reference's shape.cpp (via shapes/sphere.inl), not declared in a header
void sphere_intersect_func(const RTCIntersectFunctionNArguments *args);
void sphere_occluded_func(const RTCOccludedFunctionNArguments *args);

static uint64_t lcg_state = 0x9E3779B97F4A7C15ULL;
static double urand() {   // own deterministic generator for test inputs
    lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0);
}
static Vector3 rand_dir() {
    for (;;) {
        Vector3 v{2 * urand() - 1, 2 * urand() - 1, 2 * urand() - 1};
        double l = length(v);
        if (l > 0.1 && l <= 1) return v / l;
    }
}

static void pv(const char *k, const Vector3 &v, bool comma = true) { printf("\"%s\":[%.17g,%.17g,%.17g]%s", k, v.x, v.y, v.z, comma ? "," : ""); }
static void pv2(const char *k, const Vector2 &v, bool comma = true) { printf("\"%s\":[%.17g,%.17g]%s", k, v.x, v.y, comma ? "," : ""); }
static void pd(const char *k, double v, bool comma = true) { printf("\"%s\":%.17g%s", k, v, comma ? "," : ""); }
static void pm(const char *k, const Matrix4x4 &m, bool comma = true) {
    printf("\"%s\":[", k);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) printf("%.17g%s", m(i, j), (i == 3 && j == 3) ? "" : ",");
    printf("]%s", comma ? "," : "");
}

static void emit_pcg() {
    printf("\"pcg\":[");
    const uint64_t streams[] = {0, 1, 527, 1023, 262143, 4194303999ULL};
    for (size_t s = 0; s < sizeof(streams) / sizeof(streams[0]); s++) {
        pcg32_state st = init_pcg32(streams[s]);
        printf("%s{\"stream\":%llu,\"state\":\"%llu\",\"inc\":\"%llu\",\"u32\":[", s ? "," : "", (unsigned long long)streams[s],
               (unsigned long long)st.state, (unsigned long long)st.inc);
        pcg32_state a = st;
        for (int i = 0; i < 8; i++) printf("%u%s", next_pcg32(a), i == 7 ? "" : ",");
        printf("],\"f64\":[");
        pcg32_state b = st;
        for (int i = 0; i < 8; i++) printf("%.17g%s", next_pcg32_real<double>(b), i == 7 ? "" : ",");
        printf("]}");
    }
    printf("],");
}

static void emit_filters() {
    printf("\"filters\":[");
    struct F { int type; double param; } fs[] = {{0, 1.0}, {0, 2.5}, {1, 2.0}, {1, 3.0}, {2, 0.5}, {2, 1.25}};
    bool first = true;
    for (auto &f : fs) {
        Filter flt = f.type == 0 ? Filter(Box{f.param}) : (f.type == 1 ? Filter(Tent{f.param}) : Filter(Gaussian{f.param}));
        for (int i = 0; i < 8; i++) {
            double u0 = (i == 0) ? 0.0 : (i == 1 ? 1e-12 : (i == 2 ? 0.5 : urand())), u1 = (i == 2) ? 0.5 : urand();
            Vector2 o = sample(flt, Vector2{u0, u1});
            printf("%s{\"type\":%d,\"param\":%.17g,\"u\":[%.17g,%.17g],\"out\":[%.17g,%.17g]}", first ? "" : ",", f.type, f.param, u0, u1, o.x, o.y);
            first = false;
        }
    }
    printf("],");
}

static void emit_camera() {
    printf("\"cameras\":[");
    struct C { Vector3 pos, target, up; double fov; int w, h; int ftype; double fparam; } cs[] = {
        {Vector3{278, 273, -800}, Vector3{278, 273, -799}, Vector3{0, 1, 0}, (double)39.3077f, 512, 512, 2, 0.5},
        {Vector3{(double)3.69558f, (double)-3.46243f, (double)3.25463f}, Vector3{(double)3.04072f, (double)-2.85176f, (double)2.80939f},
         Vector3{(double)-0.317366f, (double)0.312466f, (double)0.895346f}, 37.5, 683, 512, 0, 1.0},
        {Vector3{0, 1, 5}, Vector3{0.5, 0.2, 0.0}, Vector3{0, 1, 0}, 60.0, 1280, 720, 1, 2.0}};
    for (size_t ci = 0; ci < 3; ci++) {
        C &c = cs[ci];
        Matrix4x4 to_world = look_at(c.pos, c.target, c.up);
        Filter flt = c.ftype == 0 ? Filter(Box{c.fparam}) : (c.ftype == 1 ? Filter(Tent{c.fparam}) : Filter(Gaussian{c.fparam}));
        Camera cam(to_world, c.fov, c.w, c.h, flt, -1);
        printf("%s{", ci ? "," : "");
        pv("pos", c.pos); pv("target", c.target); pv("up", c.up); pd("fov", c.fov);
        printf("\"width\":%d,\"height\":%d,\"filter_type\":%d,\"filter_param\":%.17g,", c.w, c.h, c.ftype, c.fparam);
        pm("cam_to_world", cam.cam_to_world); pm("sample_to_cam", cam.sample_to_cam);
        printf("\"rays\":[");
        for (int i = 0; i < 24; i++) {
            // screen positions as grad_path_tracing forms them: ((x+ox)+rng)/w, incl. off-image offsets
            int x = (i < 4) ? (i == 0 ? 0 : (i == 1 ? c.w - 1 : (i == 2 ? -1 : c.w))) : (int)(urand() * c.w);
            int y = (i < 4) ? (i == 0 ? 0 : (i == 1 ? c.h - 1 : (i == 2 ? 3 : -1))) : (int)(urand() * c.h);
            double rx = urand(), ry = urand();
            Vector2 sp((x + rx) / c.w, (y + ry) / c.h);
            Ray r = sample_primary(cam, sp);
            printf("%s{\"x\":%d,\"y\":%d,\"rng\":[%.17g,%.17g],\"screen\":[%.17g,%.17g],", i ? "," : "", x, y, rx, ry, sp.x, sp.y);
            pv("org", r.org); pv("dir", r.dir, false);
            printf("}");
        }
        printf("]}");
    }
    printf("],");
}

struct MatSpec { const char *name; Material m; };

static void print_tex3(const char *k, const Vector3 &v) { printf("\"%s\":[%.17g,%.17g,%.17g],", k, v.x, v.y, v.z); }

static void emit_bsdfs() {
    TexturePool pool;
    auto cs = [](double r, double g, double b) { return Texture<Spectrum>(make_constant_spectrum_texture(Vector3{r, g, b})); };
    auto cf = [](double v) { return Texture<Real>(make_constant_float_texture(v)); };
    Texture<Spectrum> checker = make_checkerboard_spectrum_texture(Vector3{0.4, 0.4, 0.4}, Vector3{0.2, 0.2, 0.2}, 8.0, 8.0, 0.0, 0.0);
    // the test rebuilds the same materials from these numbers (slot order of include/gdpt.h)
    std::vector<MatSpec> mats;
    mats.push_back({"lambertian", Lambertian{cs(0.884773640541284, 0.6999329530439845, 0.666223595587696)}});
    mats.push_back({"lambertian_checker", Lambertian{checker}});
    mats.push_back({"disneydiffuse", DisneyDiffuse{cs(0.82, 0.67, 0.16), cf(0.3), cf(0.6)}});
    mats.push_back({"disneymetal", DisneyMetal{cs(0.82, 0.67, 0.16), cf(0.2), cf(0.7)}});
    mats.push_back({"disneymetal_smooth", DisneyMetal{cs(0.9, 0.9, 0.9), cf(0.004), cf(0.0)}});
    mats.push_back({"disneyglass", DisneyGlass{cs(0.82, 0.67, 0.16), cf(0.15), cf(0.3), 1.5}});
    mats.push_back({"disneyclearcoat", DisneyClearcoat{cf(0.6)}});
    mats.push_back({"disneysheen", DisneySheen{cs(0.82, 0.67, 0.16), cf(0.4)}});
    mats.push_back({"disneybsdf", DisneyBSDF{cs(0.82, 0.67, 0.16), cf(0.5), cf(0.5), cf(0.5), cf(0.5), cf(0.1), cf(0.5), cf(0.5), cf(0.5), cf(0.5), cf(0.5), cf(0.5), 1.5}});
    mats.push_back({"disneybsdf_b", DisneyBSDF{cs(0.2, 0.5, 0.9), cf(0.1), cf(0.2), cf(0.3), cf(0.8), cf(0.45), cf(0.25), cf(0.0), cf(1.0), cf(0.3), cf(0.9), cf(0.2), 1.33}});
    mats.push_back({"roughplastic", RoughPlastic{cs(0.6, 0.3, 0.2), cs(0.9, 0.9, 0.8), cf(0.25), 1.49 / 1.000277}});
    mats.push_back({"roughplastic_checker", RoughPlastic{checker, cs(1.0, 1.0, 1.0), cf(0.004), 1.3}});
    mats.push_back({"roughdielectric", RoughDielectric{cs(0.95, 0.9, 1.0), cs(0.8, 0.9, 0.7), cf(0.3), 1.5046 / 1.000277}});
    mats.push_back({"roughdielectric_smooth", RoughDielectric{cs(1.0, 1.0, 1.0), cs(1.0, 1.0, 1.0), cf(0.02), 1.33}});
    mats.push_back({"disneybsdf_black", DisneyBSDF{cs(0.0, 0.0, 0.0), cf(0.9), cf(0.0), cf(0.0), cf(0.5), cf(0.7), cf(1.0), cf(0.9), cf(0.0), cf(0.5), cf(0.0), cf(1.0), 1.8}});
    printf("\"bsdf\":[");
    bool first = true;
    for (auto &ms : mats) {
        for (int i = 0; i < 48; i++) {
            PathVertex v;
            Vector3 gn = rand_dir();
            Vector3 sn = gn;
            if (i % 3 != 0) sn = normalize(gn + 0.35 * rand_dir());       // shading normal != geometric normal
            v.geometric_normal = (dot(gn, sn) < 0) ? -gn : gn;             // intersect() keeps them on one side
            v.shading_frame = Frame(sn);
            v.position = Vector3{0, 0, 0};
            v.uv = Vector2{urand() * 3 - 1, urand() * 3 - 1};
            v.uv_screen_size = (i % 4 == 0) ? 0.0 : urand() * 0.02;
            v.st = Vector2{0.3, 0.3};
            Vector3 dir_in = rand_dir();
            if (i % 5 != 0 && dot(dir_in, v.geometric_normal) < 0) dir_in = -dir_in;   // mostly from above
            Vector2 ruv{urand(), urand()};
            if (i == 7) ruv.x = 0.25; if (i == 8) ruv.x = 0.5; if (i == 9) ruv.x = 0.75; if (i == 10) ruv.y = 0.0;
            Real rw = urand();
            Vector3 dir_out2 = rand_dir();
            std::optional<BSDFSampleRecord> s = sample_bsdf(ms.m, dir_in, v, pool, ruv, rw);
            printf("%s{\"mat\":\"%s\",", first ? "" : ",", ms.name);
            first = false;
            pv("gn", v.geometric_normal); pv("fx", v.shading_frame.x); pv("fy", v.shading_frame.y); pv("fn", v.shading_frame.n);
            pv2("uv", v.uv); pd("uvss", v.uv_screen_size); pv("dir_in", dir_in); pv2("ruv", ruv); pd("rw", rw); pv("dir_out2", dir_out2);
            printf("\"sample_valid\":%d,", s ? 1 : 0);
            if (s) {
                pv("s_dir", s->dir_out); pd("s_eta", s->eta); pd("s_rough", s->roughness);
                Spectrum f = eval(ms.m, dir_in, s->dir_out, v, pool);
                Real p = pdf_sample_bsdf(ms.m, dir_in, s->dir_out, v, pool);
                pv("s_f", f); pd("s_pdf", p);
            }
            Spectrum f2 = eval(ms.m, dir_in, dir_out2, v, pool);
            Real p2 = pdf_sample_bsdf(ms.m, dir_in, dir_out2, v, pool);
            pv("f2", f2); pd("pdf2", p2, false);
            printf("}");
        }
    }
    printf("],");
}

static void emit_textures() {
    TexturePool pool;
    Image3 img(5, 3);   // non-power-of-two on purpose
    for (int y = 0; y < 3; y++) for (int x = 0; x < 5; x++) img(x, y) = Vector3{0.1 * x + 0.01 * y, 0.3 + 0.05 * y, 1.0 / (1 + x + y)};
    Image3 img2(16, 16);
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) img2(x, y) = Vector3{urand(), urand(), urand()};
    Texture<Spectrum> t_img = make_image_spectrum_texture("kat_img", img, pool, 2.0, 3.0, 0.25, -0.4);
    Texture<Spectrum> t_img2 = make_image_spectrum_texture("kat_img2", img2, pool, 1.0, 1.0, 0.0, 0.0);
    Texture<Spectrum> t_chk = make_checkerboard_spectrum_texture(Vector3{0.4, 0.5, 0.6}, Vector3{0.1, 0.2, 0.3}, 8.0, 4.0, 0.1, 0.2);
    Image1 img1(4, 4);
    for (int i = 0; i < 16; i++) img1(i) = urand();
    Texture<Real> t_f = make_image_float_texture("kat_f", img1, pool, 1.5, 1.5, 0.0, 0.5);
    printf("\"textures\":{");
    printf("\"img\":{\"w\":5,\"h\":3,\"uscale\":2.0,\"vscale\":3.0,\"uoffset\":0.25,\"voffset\":-0.4,\"texels\":[");
    for (int i = 0; i < 15; i++) printf("%.17g,%.17g,%.17g%s", img(i).x, img(i).y, img(i).z, i == 14 ? "" : ",");
    printf("]},\"img2\":{\"w\":16,\"h\":16,\"uscale\":1.0,\"vscale\":1.0,\"uoffset\":0.0,\"voffset\":0.0,\"texels\":[");
    for (int i = 0; i < 256; i++) printf("%.17g,%.17g,%.17g%s", img2(i).x, img2(i).y, img2(i).z, i == 255 ? "" : ",");
    printf("]},\"f\":{\"w\":4,\"h\":4,\"uscale\":1.5,\"vscale\":1.5,\"uoffset\":0.0,\"voffset\":0.5,\"texels\":[");
    for (int i = 0; i < 16; i++) printf("%.17g%s", img1(i), i == 15 ? "" : ",");
    printf("]},\"chk\":{\"c0\":[0.4,0.5,0.6],\"c1\":[0.1,0.2,0.3],\"uscale\":8.0,\"vscale\":4.0,\"uoffset\":0.1,\"voffset\":0.2},\"lookups\":[");
    bool first = true;
    for (int i = 0; i < 60; i++) {
        Vector2 uv{urand() * 4 - 2, urand() * 4 - 2};
        double fp = (i % 3 == 0) ? 0.0 : std::pow(10.0, -3.0 * urand()) * (i % 2 ? 1.0 : 0.2);
        const char *which = (i % 4 == 0) ? "img" : (i % 4 == 1 ? "img2" : (i % 4 == 2 ? "chk" : "f"));
        Vector3 o;
        if (i % 4 == 0) o = eval(t_img, uv, fp, pool);
        else if (i % 4 == 1) o = eval(t_img2, uv, fp, pool);
        else if (i % 4 == 2) o = eval(t_chk, uv, fp, pool);
        else { Real r = eval(t_f, uv, fp, pool); o = Vector3{r, r, r}; }
        printf("%s{\"tex\":\"%s\",\"uv\":[%.17g,%.17g],\"footprint\":%.17g,\"out\":[%.17g,%.17g,%.17g]}", first ? "" : ",", which, uv.x, uv.y, fp, o.x, o.y, o.z);
        first = false;
    }
    printf("]},");
}

static void emit_shading_info() {
    printf("\"shading_info\":[");
    bool first = true;
    for (int variant = 0; variant < 4; variant++) {
        TriangleMesh mesh;
        mesh.positions = {Vector3{0.1, 0.2, 0.3}, Vector3{2.0, 0.3, -0.4}, Vector3{0.5, 1.7, 0.9}, Vector3{-1.0, 1.0, 2.0}};
        mesh.indices = {Vector3i{0, 1, 2}, Vector3i{0, 2, 3}};
        if (variant >= 1) mesh.normals = compute_normal(mesh.positions, mesh.indices);
        if (variant == 2) mesh.uvs = {Vector2{0.1, 0.2}, Vector2{0.9, 0.15}, Vector2{0.4, 0.8}, Vector2{0.0, 1.0}};
        if (variant == 3) mesh.uvs = {Vector2{0.5, 0.5}, Vector2{0.5, 0.5}, Vector2{0.5, 0.5}, Vector2{0.5, 0.5}};   // degenerate
        Shape shape = mesh;
        for (int i = 0; i < 6; i++) {
            PathVertex v;
            v.primitive_id = i % 2;
            double s = urand(), t = urand();
            if (s + t > 1) { s = 1 - s; t = 1 - t; }
            v.st = Vector2{s, t};
            Vector3i idx = mesh.indices[v.primitive_id];
            Vector3 gn = normalize(cross(mesh.positions[idx[1]] - mesh.positions[idx[0]], mesh.positions[idx[2]] - mesh.positions[idx[0]]));
            v.geometric_normal = gn;
            ShadingInfo si = compute_shading_info(shape, v);
            printf("%s{\"variant\":%d,\"prim\":%d,", first ? "" : ",", variant, v.primitive_id);
            first = false;
            pv2("st", v.st); pv("gn", gn); pv2("uv", si.uv); pv("fx", si.shading_frame.x); pv("fy", si.shading_frame.y); pv("fn", si.shading_frame.n);
            pd("mean_curvature", si.mean_curvature); pd("inv_uv_size", si.inv_uv_size, false);
            printf("}");
        }
    }
    printf("],");
    // the mesh of the variants (positions, indices, computed normals) so the test can rebuild it
    TriangleMesh mesh;
    mesh.positions = {Vector3{0.1, 0.2, 0.3}, Vector3{2.0, 0.3, -0.4}, Vector3{0.5, 1.7, 0.9}, Vector3{-1.0, 1.0, 2.0}};
    mesh.indices = {Vector3i{0, 1, 2}, Vector3i{0, 2, 3}};
    std::vector<Vector3> nrm = compute_normal(mesh.positions, mesh.indices);
    printf("\"shading_mesh\":{\"positions\":[");
    for (int i = 0; i < 4; i++) printf("%.17g,%.17g,%.17g%s", mesh.positions[i].x, mesh.positions[i].y, mesh.positions[i].z, i == 3 ? "" : ",");
    printf("],\"indices\":[0,1,2,0,2,3],\"normals\":[");
    for (int i = 0; i < 4; i++) printf("%.17g,%.17g,%.17g%s", nrm[i].x, nrm[i].y, nrm[i].z, i == 3 ? "" : ",");
    printf("]},");
    // sphere shading info (src/shapes/sphere.inl:243-268)
    printf("\"sphere_shading\":[");
    Sphere sp{{}, Vector3{1.0, -2.0, 0.5}, 2.5};
    Shape sshape = sp;
    for (int i = 0; i < 6; i++) {
        PathVertex v;
        v.st = Vector2{(double)(float)(urand() - 0.5), (double)(float)urand()};
        v.geometric_normal = rand_dir();
        ShadingInfo si = compute_shading_info(sshape, v);
        printf("%s{", i ? "," : "");
        pv2("st", v.st); pv("gn", v.geometric_normal); pv2("uv", si.uv); pv("fx", si.shading_frame.x); pv("fy", si.shading_frame.y);
        pv("fn", si.shading_frame.n); pd("mean_curvature", si.mean_curvature); pd("inv_uv_size", si.inv_uv_size, false);
        printf("}");
    }
    printf("],");
}

static void emit_sphere_hits() {
    printf("\"sphere_hits\":{\"center\":[1.0,-2.0,0.5],\"radius\":2.5,\"cases\":[");
    Sphere sp{{}, Vector3{1.0, -2.0, 0.5}, 2.5};
    RTCRayQueryContext ctx;
    rtcInitRayQueryContext(&ctx);
    for (int i = 0; i < 24; i++) {
        RTCRayHit rh;
        Vector3 o = (i % 4 == 3) ? Vector3{1.0 + 0.5 * (urand() - 0.5), -2.0 + 0.5 * (urand() - 0.5), 0.5} : Vector3{10 * (urand() - 0.5), 10 * (urand() - 0.5), 10 * (urand() - 0.5)};
        Vector3 d = (i % 3 == 0) ? normalize(sp.position - o + 2.0 * rand_dir()) : rand_dir();
        if (i % 4 == 3) d = rand_dir();
        rh.ray.org_x = (float)o.x; rh.ray.org_y = (float)o.y; rh.ray.org_z = (float)o.z;
        rh.ray.dir_x = (float)d.x; rh.ray.dir_y = (float)d.y; rh.ray.dir_z = (float)d.z;
        rh.ray.tnear = (i % 5 == 0) ? 0.01f : 0.0f;
        rh.ray.tfar = (i == 11) ? 3.0f : std::numeric_limits<float>::infinity();
        rh.ray.time = 0; rh.ray.mask = (unsigned)-1; rh.ray.id = 0; rh.ray.flags = 0;
        rh.hit.geomID = RTC_INVALID_GEOMETRY_ID; rh.hit.primID = RTC_INVALID_GEOMETRY_ID; rh.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
        rh.hit.Ng_x = rh.hit.Ng_y = rh.hit.Ng_z = 0; rh.hit.u = rh.hit.v = 0;
        float tnear0 = rh.ray.tnear, tfar0 = rh.ray.tfar;
        int valid = -1;
        RTCIntersectFunctionNArguments args;
        args.valid = &valid; args.geometryUserPtr = (void *)&sp; args.primID = 0; args.context = &ctx;
        args.rayhit = (RTCRayHitN *)&rh; args.N = 1; args.geomID = 7;
        sphere_intersect_func(&args);
        int hit = rh.hit.geomID == 7;
        printf("%s{\"org\":[%.9g,%.9g,%.9g],\"dir\":[%.9g,%.9g,%.9g],\"tnear\":%.9g,\"tfar\":%s,\"hit\":%d", i ? "," : "",
               rh.ray.org_x, rh.ray.org_y, rh.ray.org_z, rh.ray.dir_x, rh.ray.dir_y, rh.ray.dir_z, tnear0,
               std::isinf(tfar0) ? "\"inf\"" : "3.0", hit);
        if (hit) printf(",\"t\":%.9g,\"u\":%.9g,\"v\":%.9g,\"ng\":[%.9g,%.9g,%.9g]", rh.ray.tfar, rh.hit.u, rh.hit.v, rh.hit.Ng_x, rh.hit.Ng_y, rh.hit.Ng_z);
        printf("}");
    }
    printf("]},");
}

static void emit_obj(const std::string &scene_dir) {
    printf("\"obj\":[");
    const char *files[] = {"cbox_luminaire.obj", "cbox_floor.obj", "cbox_ceiling.obj", "cbox_back.obj", "cbox_greenwall.obj",
                           "cbox_redwall.obj", "cbox_smallbox.obj", "cbox_largebox.obj"};
    for (int f = 0; f < 8; f++) {
        Matrix4x4 to_world = (f == 0) ? translate(Vector3{0.0, (double)-0.5f, 0.0}) : Matrix4x4::identity();
        if (f == 7) to_world = rotate(30.0, Vector3{0.2, 1.0, 0.1}) * scale(Vector3{1.5, 1.0, 0.5});   // exercise normals under a general transform
        TriangleMesh mesh = parse_obj(scene_dir + "/meshes/" + files[f], to_world);
        bool had_normals = mesh.normals.size() > 0;
        std::vector<Vector3> nrm = had_normals ? mesh.normals : compute_normal(mesh.positions, mesh.indices);
        printf("%s{\"file\":\"%s\",\"variant\":%d,\"had_normals\":%d,", f ? "," : "", files[f], f == 0 ? 1 : (f == 7 ? 2 : 0), had_normals ? 1 : 0);
        pm("to_world", to_world);
        printf("\"positions\":[");
        for (size_t i = 0; i < mesh.positions.size(); i++) printf("%.17g,%.17g,%.17g%s", mesh.positions[i].x, mesh.positions[i].y, mesh.positions[i].z, i + 1 == mesh.positions.size() ? "" : ",");
        printf("],\"indices\":[");
        for (size_t i = 0; i < mesh.indices.size(); i++) printf("%d,%d,%d%s", mesh.indices[i][0], mesh.indices[i][1], mesh.indices[i][2], i + 1 == mesh.indices.size() ? "" : ",");
        printf("],\"normals\":[");
        for (size_t i = 0; i < nrm.size(); i++) printf("%.17g,%.17g,%.17g%s", nrm[i].x, nrm[i].y, nrm[i].z, i + 1 == nrm.size() ? "" : ",");
        printf("],\"num_uvs\":%d}", (int)mesh.uvs.size());
    }
    printf("],");
}

// Emitter sampling of Integrator::Path: table distributions (src/table_dist.cpp),
// sample_point_on_shape / pdf_point_on_shape / surface_area for a triangle mesh (the cbox luminaire under its scene
// transform, and the large box with vertex normals under a general transform) and a sphere, and the sphere's
// occlusion callback (src/shapes/sphere.inl:108-146).
static void emit_light_sampling(const std::string &scene_dir) {
    printf("\"light_sampling\":{");
    {   // TableDist1D
        std::vector<Real> f = {0.5, 0.0, 2.25, 1.0, 0.0, 3.5};
        TableDist1D t = make_table_dist_1d(f);
        printf("\"table\":{\"f\":[0.5,0.0,2.25,1.0,0.0,3.5],\"pmf\":[");
        for (size_t i = 0; i < t.pmf.size(); i++) printf("%s%.17g", i ? "," : "", t.pmf[i]);
        printf("],\"cdf\":[");
        for (size_t i = 0; i < t.cdf.size(); i++) printf("%s%.17g", i ? "," : "", t.cdf[i]);
        printf("],\"samples\":[");
        const double us[] = {0.0, 0.0689655172413793, 0.07, 0.3, 0.379310344827586, 0.5, 0.5172413793103449, 0.52, 0.9999999999, 0.25};
        for (int i = 0; i < 10; i++) printf("%s[%.17g,%d]", i ? "," : "", us[i], sample(t, us[i]));
        printf("]},");
    }
    {   // TableDist2D (the environment map's sampling table), incl. a black row and a single bright texel
        const int w = 6, h = 4;
        std::vector<Real> f = {0.2, 0.5, 0.0, 1.5, 0.25, 0.75,
                               0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                               3.0, 0.125, 0.5, 0.5, 2.0, 0.0625,
                               0.0, 0.0, 9.0, 0.0, 0.0, 0.0};
        TableDist2D t = make_table_dist_2d(f, w, h);
        printf("\"table2d\":{\"width\":6,\"height\":4,\"f\":[");
        for (size_t i = 0; i < f.size(); i++) printf("%s%.17g", i ? "," : "", f[i]);
        printf("],\"total\":%.17g,\"samples\":[", t.total_values);
        for (int i = 0; i < 40; i++) {
            Vector2 r{urand(), urand()};
            if (i == 0) r = Vector2{0.0, 0.0};
            if (i == 1) r = Vector2{0.999999999, 0.999999999};
            if (i == 2) r = Vector2{0.5, 0.2};
            Vector2 uv = sample(t, r);
            printf("%s{\"rnd\":[%.17g,%.17g],\"uv\":[%.17g,%.17g],\"pdf\":%.17g}", i ? "," : "", r.x, r.y, uv.x, uv.y, pdf(t, uv));
        }
        printf("]},");
    }
    printf("\"meshes\":[");
    for (int m = 0; m < 2; m++) {
        Matrix4x4 to_world = (m == 0) ? translate(Vector3{0.0, (double)-0.5f, 0.0}) : rotate(30.0, Vector3{0.2, 1.0, 0.1}) * scale(Vector3{1.5, 1.0, 0.5});
        const char *file = m == 0 ? "cbox_luminaire.obj" : "cbox_largebox.obj";
        TriangleMesh mesh = parse_obj(scene_dir + "/meshes/" + file, to_world);
        if (m == 1) mesh.normals = compute_normal(mesh.positions, mesh.indices);      // exercise the shading-side flip
        Shape shape = mesh;
        init_sampling_dist(shape);
        printf("%s{\"file\":\"%s\",\"variant\":%d,\"with_normals\":%d,\"area\":%.17g,\"cases\":[", m ? "," : "", file, m == 0 ? 1 : 2, m, surface_area(shape));
        for (int i = 0; i < 16; i++) {
            Vector3 ref = Vector3{600 * urand(), 600 * urand(), 600 * urand()};
            Vector2 uv{urand(), urand()};
            if (i == 3) uv = Vector2{0.0, 0.0};
            if (i == 4) uv = Vector2{1.0, 1.0};
            Real w = urand();
            PointAndNormal pn = sample_point_on_shape(shape, ref, uv, w);
            Real pdf = pdf_point_on_shape(shape, pn, ref);
            printf("%s{\"ref\":[%.17g,%.17g,%.17g],\"uv\":[%.17g,%.17g],\"w\":%.17g,\"position\":[%.17g,%.17g,%.17g],\"normal\":[%.17g,%.17g,%.17g],\"pdf\":%.17g}",
                   i ? "," : "", ref.x, ref.y, ref.z, uv.x, uv.y, w, pn.position.x, pn.position.y, pn.position.z, pn.normal.x, pn.normal.y, pn.normal.z, pdf);
        }
        printf("]}");
    }
    printf("],\"sphere\":{\"center\":[1.0,-2.0,0.5],\"radius\":2.5,");
    {
        Sphere sp{{}, Vector3{1.0, -2.0, 0.5}, 2.5};
        Shape shape = sp;
        printf("\"area\":%.17g,\"cases\":[", surface_area(shape));
        for (int i = 0; i < 20; i++) {
            Vector3 ref = (i % 4 == 3) ? Vector3{1.0 + 2.0 * (urand() - 0.5), -2.0 + 2.0 * (urand() - 0.5), 0.5 + 2.0 * (urand() - 0.5)}   // inside
                                       : Vector3{1.0, -2.0, 0.5} + (3.0 + 20.0 * urand()) * rand_dir();
            Vector2 uv{urand(), urand()};
            if (i == 5) uv = Vector2{0.0, 0.25};
            if (i == 6) uv = Vector2{1.0, 0.75};
            PointAndNormal pn = sample_point_on_shape(shape, ref, uv, urand());
            Real pdf = pdf_point_on_shape(shape, pn, ref);
            printf("%s{\"ref\":[%.17g,%.17g,%.17g],\"uv\":[%.17g,%.17g],\"position\":[%.17g,%.17g,%.17g],\"normal\":[%.17g,%.17g,%.17g],\"pdf\":%.17g}",
                   i ? "," : "", ref.x, ref.y, ref.z, uv.x, uv.y, pn.position.x, pn.position.y, pn.position.z, pn.normal.x, pn.normal.y, pn.normal.z, pdf);
        }
        printf("],\"occluded\":[");
        for (int i = 0; i < 16; i++) {
            RTCRay ray;
            Vector3 o = Vector3{10 * (urand() - 0.5), 10 * (urand() - 0.5), 10 * (urand() - 0.5)};
            Vector3 d = (i % 2 == 0) ? normalize(sp.position - o + 2.0 * rand_dir()) : rand_dir();
            ray.org_x = (float)o.x; ray.org_y = (float)o.y; ray.org_z = (float)o.z;
            ray.dir_x = (float)d.x; ray.dir_y = (float)d.y; ray.dir_z = (float)d.z;
            ray.tnear = (i % 5 == 0) ? 0.01f : 0.0f;
            ray.tfar = (i % 3 == 0) ? (float)(0.5 + 6.0 * urand()) : std::numeric_limits<float>::infinity();
            ray.time = 0; ray.mask = (unsigned)-1; ray.id = 0; ray.flags = 0;
            float tfar0 = ray.tfar;
            int valid = -1;
            RTCOccludedFunctionNArguments args;
            args.valid = &valid; args.geometryUserPtr = (void *)&sp; args.primID = 0; args.context = nullptr;
            args.ray = (RTCRayN *)&ray; args.N = 1; args.geomID = 7;
            sphere_occluded_func(&args);
            printf("%s{\"org\":[%.9g,%.9g,%.9g],\"dir\":[%.9g,%.9g,%.9g],\"tnear\":%.9g,\"tfar\":%s%.9g%s,\"occluded\":%d}", i ? "," : "",
                   ray.org_x, ray.org_y, ray.org_z, ray.dir_x, ray.dir_y, ray.dir_z, ray.tnear,
                   std::isinf(tfar0) ? "\"" : "", std::isinf(tfar0) ? 0.0 : tfar0, std::isinf(tfar0) ? "inf\"" : "", ray.tfar < 0 ? 1 : 0);
        }
        printf("]}},");
    }
}

// spectra file: one spectrum per line, "w0 v0 w1 v1 ..." (already rounded through fp32 by the generator script)
static void emit_spectra(const std::string &path) {
    printf("\"spectra\":[");
    std::ifstream ifs(path);
    std::string line;
    bool first = true;
    while (std::getline(ifs, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        std::vector<std::pair<Real, Real>> data;
        double w, v;
        while (ss >> w >> v) data.push_back({w, v});
        Vector3 xyz = integrate_XYZ(data);
        Vector3 rgb = XYZ_to_RGB(xyz);
        printf("%s{\"n\":%d,\"first\":[%.17g,%.17g],\"xyz\":[%.17g,%.17g,%.17g],\"rgb\":[%.17g,%.17g,%.17g]}", first ? "" : ",", (int)data.size(),
               data[0].first, data[0].second, xyz.x, xyz.y, xyz.z, rgb.x, rgb.y, rgb.z);
        first = false;
    }
    printf("],");
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_kat <cbox scene dir> <spectra file>\n"); return 2; }
    printf("{");
    emit_pcg();
    emit_filters();
    emit_camera();
    emit_bsdfs();
    emit_textures();
    emit_shading_info();
    emit_sphere_hits();
    emit_obj(argv[1]);
    emit_light_sampling(argv[1]);
    emit_spectra(argv[2]);
    printf("\"generator\":\"oracle/ref_kat.cpp linked against the reference sources (see oracle/Makefile)\"}\n");
    return 0;
}
