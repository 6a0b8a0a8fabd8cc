// device_bsdf.h — textures + the BSDF trio (eval = f*|cos|, pdf, sample) in fp64 for the HIP kernels.
// Restates src/material.cpp:90-119 and src/materials/{lambertian,disney_*}.inl, src/microfacet.h,
// src/texture.h:112-159, src/mipmap.h:51-88 — including the quirks listed in SURVEY.md §8(a) Q5
// (unclamped roughness in the metal pdf, |n.out| in the clearcoat pdf, isotropic VNDF for glass,
// fixed 0.25/0.5/0.75 lobe thresholds, doubly flipped eta of the inner glass lobe, sheen excluded
// from the pdf normalisation). pow(x,2)/pow(x,0.5)/pow(x,5) are evaluated as products/sqrt.
#pragma once
#include "../../../include/gdpt.h"
#include "../device_scene.h"
#include "device_math.h"

namespace gd {

struct Vertex {            // PathVertex (src/intersection.h:15-37), fields the hot path reads
    D3 position, gn;       // gn: geometric normal flipped to the shading side
    Frame frame;
    D2 uv;
    double uv_screen_size;
    int material_id, light_id;
};

struct BsdfSample { D3 dir_out; double eta, roughness; };

// ---- textures -----------------------------------------------------------------------------------
GD D3 mip_lookup_level(const DevSceneView &sv, const DevImage &im, double u, double v, int level) {
    int w = im.width[level], h = im.height[level];
    const double *img = sv.texels + im.offset[level];
    u = u * w - 0.5;
    v = v * h - 0.5;
    int ufi = modulo_i(int(u), w), vfi = modulo_i(int(v), h);
    int uci = modulo_i(ufi + 1, w), vci = modulo_i(vfi + 1, h);
    double u_off = u - ufi, v_off = v - vfi;
    D3 ff, fc, cf, cc;
    if (im.channels == 1) {
        ff = splat(img[(size_t)vfi * w + ufi]); fc = splat(img[(size_t)vci * w + ufi]);
        cf = splat(img[(size_t)vfi * w + uci]); cc = splat(img[(size_t)vci * w + uci]);
    } else {
        const double *p;
        p = img + ((size_t)vfi * w + ufi) * 3; ff = mk(p[0], p[1], p[2]);
        p = img + ((size_t)vci * w + ufi) * 3; fc = mk(p[0], p[1], p[2]);
        p = img + ((size_t)vfi * w + uci) * 3; cf = mk(p[0], p[1], p[2]);
        p = img + ((size_t)vci * w + uci) * 3; cc = mk(p[0], p[1], p[2]);
    }
    return ff * (1 - u_off) * (1 - v_off) + fc * (1 - u_off) * v_off + cf * u_off * (1 - v_off) + cc * u_off * v_off;
}
GD D3 mip_lookup(const DevSceneView &sv, const DevImage &im, double u, double v, double level) {
    int n = im.num_levels;
    if (level <= 0) return mip_lookup_level(sv, im, u, v, 0);
    if (level < double(n - 1)) {
        int fl = min(max((int)floor(level), 0), n - 1);
        int cl = min(max(fl + 1, 0), n - 1);
        double off = level - fl;
        return mip_lookup_level(sv, im, u, v, fl) * (1 - off) + mip_lookup_level(sv, im, u, v, cl) * off;
    }
    return mip_lookup_level(sv, im, u, v, n - 1);
}
GD D3 tex3(const DevSceneView &sv, const GdptTexture &t, const Vertex &vx) {
    if (t.type == GDPT_TEX_CONSTANT) return mk(t.v0[0], t.v0[1], t.v0[2]);
    double lu = modulo_d(vx.uv.x * t.uscale + t.uoffset, 1.0), lv = modulo_d(vx.uv.y * t.vscale + t.voffset, 1.0);
    if (t.type == GDPT_TEX_IMAGE) {
        const DevImage &im = sv.images[t.image_id];
        double scaled = fmax((double)im.width[0], (double)im.height[0]) * fmax(t.uscale, t.vscale) * vx.uv_screen_size;
        double level = log2(fmax(scaled, (double)1e-8f));
        return mip_lookup(sv, im, lu, lv, level);
    }
    int x = 2 * modulo_i((int)(lu * 2), 2) - 1, y = 2 * modulo_i((int)(lv * 2), 2) - 1;
    if (x * y == 1) return mk(t.v0[0], t.v0[1], t.v0[2]);
    return mk(t.v1[0], t.v1[1], t.v1[2]);
}
GD double tex1(const DevSceneView &sv, const GdptTexture &t, const Vertex &vx) {
    if (t.type == GDPT_TEX_CONSTANT) return t.v0[0];
    return tex3(sv, t, vx).x;
}

// (Measured and dropped, round 3: the general switch calling an out-of-line lookup for non-constant textures — pointers into
// global memory and scalars in, three doubles out, the calling pattern of sphere_hit. The two-sided lane machine shrank from
// 100 000 to 25 000 instructions and compiled five times faster, and rendered disney_glass 24 %, DisneyBSDF 16 % and
// disney_metal 15 % SLOWER (same-box A/B, profiles/r03_ab_texture_call.txt): the checkerboard floor of those scenes takes the
// call on every hit, and a call in the middle of a 256-VGPR step parks the live state in scratch. Code size is not what binds
// these kernels.)
// ---- helpers ------------------------------------------------------------------------------------
GD D3 sample_cos_hemisphere(D2 r) { // src/material.cpp:4-11
    // phi = 2*pi*u: sincospi(2u) skips the large-argument reduction of sincos (the two differ by the rounding of phi)
    double tmp = sqrt(clamp01(1 - r.y));
    double s, c;
    sincospi(2.0 * r.x, &s, &c);
    return mk(c * tmp, s * tmp, sqrt(clamp01(r.y)));
}
GD double fresnel_dielectric2(double n_dot_i, double n_dot_t, double eta) { // src/microfacet.h:34-40
    double rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
    double rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
    return (rs * rs + rp * rp) / 2;
}
GD double fresnel_dielectric(double n_dot_i, double eta) { // src/microfacet.h:47-56
    double n_dot_t_sq = 1 - (1 - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0) return 1;
    return fresnel_dielectric2(fabs(n_dot_i), sqrt(n_dot_t_sq), eta);
}
GD D3 sample_visible_normals(D3 local_in, double ax, double ay, D2 rnd) { // src/microfacet.h:96-161
    bool flip = local_in.z < 0;
    if (flip) local_in = -local_in;
    D3 hemi = normalize(mk(ax * local_in.x, ay * local_in.y, local_in.z));
    double r = sqrt(rnd.x);
    double sp, cp;
    sincospi(2.0 * rnd.y, &sp, &cp);
    double t1 = r * cp, t2 = r * sp;
    double s = (1 + hemi.z) / 2;
    t2 = (1 - s) * sqrt(1 - t1 * t1) + s * t2;
    D3 disk = mk(t1, t2, sqrt(fmax(0.0, 1 - t1 * t1 - t2 * t2)));
    Frame hf = make_frame(hemi);
    D3 hn = to_world(hf, disk);
    D3 res = normalize(mk(ax * hn.x, ay * hn.y, fmax(0.0, hn.z)));
    return flip ? -res : res;
}
GD D3 sample_clearcoat_normal(double alpha, D2 rnd) { // src/microfacet.h:164-177
    double a2 = alpha * alpha;
    double pw = pow(a2, 1 - rnd.x);
    double sin_e = sqrt((pw - a2) / (1 - a2));
    double cos_e = sqrt((1 - pw) / (1 - a2));
    double s, c;
    sincospi(2.0 * rnd.y, &s, &c);
    return normalize(mk(sin_e * c, sin_e * s, cos_e));
}

GD bool below(const Vertex &v, D3 d) { return dot(v.gn, d) < 0; }
GD Frame oriented_frame(const Vertex &v, D3 in) { return (dot(v.frame.n, in) < 0) ? neg(v.frame) : v.frame; }
GD Frame oriented_frame_2s(const Vertex &v, D3 in) { return (dot(v.frame.n, in) * dot(v.gn, in) < 0) ? neg(v.frame) : v.frame; }

struct Ctx { const DevSceneView &sv; const Vertex &v; };
GD D3 T3(const Ctx &c, const GdptTexture &t) { return tex3(c.sv, t, c.v); }
GD double T1(const Ctx &c, const GdptTexture &t) { return tex1(c.sv, t, c.v); }

// ---- cosine lobes (Lambertian / pdf+sample of DisneyDiffuse, DisneySheen) -------------------------
GD double cos_pdf(const Ctx &c, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return 0;
    Frame f = oriented_frame(c.v, in);
    return fmax(dot(f.n, out), 0.0) / kPi;
}
GD bool cos_sample(const Ctx &c, D3 in, D2 ruv, double roughness, BsdfSample &s) {
    if (below(c.v, in)) return false;
    Frame f = oriented_frame(c.v, in);
    s.dir_out = to_world(f, sample_cos_hemisphere(ruv)); s.eta = 0; s.roughness = roughness;
    return true;
}
GD D3 lambert_eval(const Ctx &c, const GdptTexture &refl, D3 in, D3 out) { // src/materials/lambertian.inl:1-17
    if (below(c.v, in) || below(c.v, out)) return splat(0);
    Frame f = oriented_frame(c.v, in);
    return fmax(dot(f.n, out), 0.0) * T3(c, refl) / kPi;
}

// ---- DisneyDiffuse, src/materials/disney_diffuse.inl ----------------------------------------------
GD D3 dd_eval(const Ctx &c, const GdptTexture &base, const GdptTexture &rough, const GdptTexture &subs, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return splat(0);
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_in = dot(f.n, in), n_out = dot(f.n, out), h_out = dot(h, out);
    double roughness = fmin(fmax(T1(c, rough), 0.01), 1.0);
    D3 bc = T3(c, base);
    double ho2 = sqr(fabs(h_out));
    double f_d_90 = 0.5 + 2 * roughness * ho2;
    double p5o = pow5(1 - fabs(n_out)), p5i = pow5(1 - fabs(n_in));
    double f_d_out = 1.0 + (f_d_90 - 1.0) * p5o, f_d_in = 1.0 + (f_d_90 - 1.0) * p5i;
    D3 f_base = (bc * f_d_in * f_d_out * fabs(n_out)) / kPi;
    double f_ss_90 = roughness * ho2;
    double f_ss_in = 1.0 + (f_ss_90 - 1.0) * p5i, f_ss_out = 1.0 + (f_ss_90 - 1.0) * p5o;
    double inner = (f_ss_in * f_ss_out) * (1 / (fabs(n_in) + fabs(n_out)) - 0.5) + 0.5;
    D3 f_ss = ((1.25 * bc) / kPi) * inner * fabs(n_out);
    double sv = T1(c, subs);
    return (1 - sv) * f_base + sv * f_ss;
}
GD bool dd_sample(const Ctx &c, const GdptTexture &rough, D3 in, D2 ruv, BsdfSample &s) {
    if (below(c.v, in)) return false;
    return cos_sample(c, in, ruv, fmin(fmax(T1(c, rough), 0.01), 1.0), s);
}

// ---- DisneyMetal, src/materials/disney_metal.inl --------------------------------------------------
GD void aniso_alpha(double roughness, double anisotropic, double &ax, double &ay) {
    double aspect = sqrt(1 - 0.9 * anisotropic);
    double r2 = roughness * roughness;
    ax = fmax(0.0001, r2 / aspect); ay = fmax(0.0001, r2 * aspect);
}
GD double ggx_aniso_D(D3 hl, double ax, double ay) {
    double t = sqr(hl.x / ax) + sqr(hl.y / ay) + sqr(hl.z);
    return 1 / (kPi * ax * ay * t * t);
}
GD double smith_lambda_term(D3 l, double ax, double ay) { // (sqrt(1 + (x^2 ax^2 + y^2 ay^2)/z^2) - 1)/2
    double in = (sqr(l.x * ax) + sqr(l.y * ay)) / sqr(l.z);
    return (sqrt(1 + in) - 1) / 2;
}
GD D3 dm_eval(const Ctx &c, D3 bc, const GdptTexture &rough, const GdptTexture &aniso, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return splat(0);
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_in = dot(f.n, in), h_out = dot(h, out);
    double roughness = fmin(fmax(T1(c, rough), 0.01), 1.0), anisotropic = T1(c, aniso);
    D3 f_m = bc + (splat(1.0) - bc) * pow5(1.0 - fabs(h_out));
    double ax, ay; aniso_alpha(roughness, anisotropic, ax, ay);
    double D = ggx_aniso_D(to_local(f, h), ax, ay);
    double G = (1 / (1 + smith_lambda_term(to_local(f, in), ax, ay))) * (1 / (1 + smith_lambda_term(to_local(f, out), ax, ay)));
    return (f_m * D * G) / (4 * fabs(n_in));
}
GD double dm_pdf(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return 0;
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_in = dot(f.n, in);
    double roughness = T1(c, rough), anisotropic = T1(c, aniso);  // roughness NOT clamped (disney_metal.inl:107-125)
    double ax, ay; aniso_alpha(roughness, anisotropic, ax, ay);
    double D = ggx_aniso_D(to_local(f, h), ax, ay);
    double G = 1 / (1 + smith_lambda_term(to_local(f, in), ax, ay));
    return (G * D) / (4 * fabs(n_in));
}
GD bool dm_sample(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, D3 in, D2 ruv, BsdfSample &s) {
    if (below(c.v, in)) return false;
    Frame f = oriented_frame(c.v, in);
    double roughness = fmin(fmax(T1(c, rough), 0.01), 1.0), anisotropic = T1(c, aniso);
    double ax, ay; aniso_alpha(roughness, anisotropic, ax, ay);
    D3 h = to_world(f, sample_visible_normals(to_local(f, in), ax, ay, ruv));
    s.dir_out = normalize(-in + 2 * dot(in, h) * h); s.eta = 0; s.roughness = roughness;
    return true;
}

// ---- DisneyClearcoat, src/materials/disney_clearcoat.inl ------------------------------------------
GD double cc_D(double alpha_g, double hz) {
    double a2 = alpha_g * alpha_g;
    return (a2 - 1) / (kPi * log(a2) * (1 + (a2 - 1) * (hz * hz)));
}
GD double cc_alpha(const Ctx &c, const GdptTexture &gloss) { double g = T1(c, gloss); return (1 - g) * 0.1 + g * 0.001; }
GD D3 cc_eval(const Ctx &c, const GdptTexture &gloss, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return splat(0);
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_in = dot(f.n, in), h_out = dot(h, out);
    const double r_0 = (0.5 * 0.5) / (2.5 * 2.5);
    double f_c = r_0 + (1 - r_0) * pow5(1 - fabs(h_out));
    double d_c = cc_D(cc_alpha(c, gloss), to_local(f, h).z);
    double g_c = (1 / (1 + smith_lambda_term(to_local(f, in), 0.25, 0.25))) * (1 / (1 + smith_lambda_term(to_local(f, out), 0.25, 0.25)));
    return splat((f_c * d_c * g_c) / (4 * fabs(n_in)));
}
GD double cc_pdf(const Ctx &c, const GdptTexture &gloss, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return 0;
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_out = dot(f.n, out), n_h = dot(f.n, h);
    double d_c = cc_D(cc_alpha(c, gloss), to_local(f, h).z);
    return (d_c * fabs(n_h)) / (4 * fabs(n_out));               // sic: |n.out| (disney_clearcoat.inl:64)
}
GD bool cc_sample(const Ctx &c, const GdptTexture &gloss, D3 in, D2 ruv, BsdfSample &s) {
    if (below(c.v, in)) return false;
    Frame f = oriented_frame(c.v, in);
    double alpha_g = cc_alpha(c, gloss);
    D3 h = to_world(f, sample_clearcoat_normal(alpha_g, ruv));
    s.dir_out = normalize(-in + 2 * dot(in, h) * h); s.eta = 0; s.roughness = alpha_g;
    return true;
}

// ---- DisneySheen, src/materials/disney_sheen.inl --------------------------------------------------
GD D3 sh_eval(const Ctx &c, const GdptTexture &base, const GdptTexture &tint, D3 in, D3 out) {
    if (below(c.v, in) || below(c.v, out)) return splat(0);
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double h_out = dot(h, out), n_out = dot(f.n, out);
    D3 bc = T3(c, base);
    double st = T1(c, tint);
    double lum = luminance(bc);
    D3 c_tint = (lum > 0) ? bc / lum : splat(1.0);
    D3 c_sheen = splat(1.0 - st) + st * c_tint;
    return c_sheen * pow5(1 - fabs(h_out)) * fabs(n_out);
}

// ---- DisneyGlass, src/materials/disney_glass.inl --------------------------------------------------
struct GlassTerms { Frame f; D3 h; double eta, F, d_m, g_in, g_out, h_dot_in; bool reflect; };
GD GlassTerms glass_terms(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, double bsdf_eta, D3 in, D3 out) {
    GlassTerms g;
    double gn_in = dot(c.v.gn, in);
    g.reflect = gn_in * dot(c.v.gn, out) > 0;
    g.f = oriented_frame_2s(c.v, in);
    g.eta = gn_in > 0 ? bsdf_eta : 1 / bsdf_eta;
    g.h = g.reflect ? normalize(in + out) : normalize(in + out * g.eta);
    if (dot(g.h, g.f.n) < 0) g.h = -g.h;
    double roughness = fmin(fmax(T1(c, rough), 0.01), 1.0), anisotropic = T1(c, aniso);
    g.h_dot_in = dot(g.h, in);
    g.F = fresnel_dielectric(g.h_dot_in, g.eta);
    double ax, ay; aniso_alpha(roughness, anisotropic, ax, ay);
    D3 pv = to_local(g.f, g.h);
    double t = sqr(pv.x) / sqr(ax) + sqr(pv.y) / sqr(ay) + sqr(pv.z);
    g.d_m = 1 / (kPi * ax * ay * t * t);
    g.g_in = 1 / (1 + smith_lambda_term(to_local(g.f, in), ax, ay));
    g.g_out = 1 / (1 + smith_lambda_term(to_local(g.f, out), ax, ay));
    return g;
}
GD D3 dg_eval(const Ctx &c, const GdptTexture &base, const GdptTexture &rough, const GdptTexture &aniso, double eta, D3 in, D3 out) {
    D3 bc = T3(c, base);
    GlassTerms g = glass_terms(c, rough, aniso, eta, in, out);
    double g_m = g.g_in * g.g_out;
    if (g.reflect) return bc * (g.F * g.d_m * g_m) / (4 * fabs(dot(g.f.n, in)));
    double h_dot_out = dot(g.h, out);
    double sd = g.h_dot_in + g.eta * h_dot_out;
    D3 csq = mk(sqrt(bc.x), sqrt(bc.y), sqrt(bc.z));
    return csq * ((1 - g.F) * g.d_m * g_m * fabs(h_dot_out * g.h_dot_in)) / (fabs(dot(g.f.n, in)) * sd * sd);
}
GD double dg_pdf(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, double eta, D3 in, D3 out) {
    GlassTerms g = glass_terms(c, rough, aniso, eta, in, out);
    if (g.reflect) return (g.F * g.d_m * g.g_in) / (4 * fabs(dot(g.f.n, in)));
    double h_dot_out = dot(g.h, out);
    double sd = g.h_dot_in + g.eta * h_dot_out;
    return ((1 - g.F) * g.d_m * g.g_in * fabs(h_dot_out * g.h_dot_in)) / (fabs(dot(g.f.n, in)) * sd * sd);
}
// eval and pdf of one direction pair share everything but the base colour and the second masking term: evaluated together
// (the callers always want both) the half vector, Fresnel term, D and G1 are formed once. Every value is produced by the
// expression that dg_eval / dg_pdf use for it, so the two results are bit-identical to the separate calls.
GD void dg_eval_pdf(const Ctx &c, const GdptTexture &base, const GdptTexture &rough, const GdptTexture &aniso, double eta, D3 in, D3 out, D3 &f, double &pdf) {
    D3 bc = T3(c, base);
    GlassTerms g = glass_terms(c, rough, aniso, eta, in, out);
    double g_m = g.g_in * g.g_out;
    if (g.reflect) {
        f = bc * (g.F * g.d_m * g_m) / (4 * fabs(dot(g.f.n, in)));
        pdf = (g.F * g.d_m * g.g_in) / (4 * fabs(dot(g.f.n, in)));
        return;
    }
    double h_dot_out = dot(g.h, out);
    double sd = g.h_dot_in + g.eta * h_dot_out;
    D3 csq = mk(sqrt(bc.x), sqrt(bc.y), sqrt(bc.z));
    f = csq * ((1 - g.F) * g.d_m * g_m * fabs(h_dot_out * g.h_dot_in)) / (fabs(dot(g.f.n, in)) * sd * sd);
    pdf = ((1 - g.F) * g.d_m * g.g_in * fabs(h_dot_out * g.h_dot_in)) / (fabs(dot(g.f.n, in)) * sd * sd);
}
GD bool dg_sample(const Ctx &c, const GdptTexture &rough, double bsdf_eta, D3 in, D2 ruv, double rw, BsdfSample &s) {
    double eta = dot(c.v.gn, in) > 0 ? bsdf_eta : 1 / bsdf_eta;
    Frame f = oriented_frame_2s(c.v, in);
    double roughness = fmin(fmax(T1(c, rough), 0.01), 1.0);
    double alpha = roughness * roughness;
    D3 h = to_world(f, sample_visible_normals(to_local(f, in), alpha, alpha, ruv)); // isotropic (disney_glass.inl:195-198)
    if (dot(h, f.n) < 0) h = -h;
    double h_dot_in = dot(h, in);
    double F = fresnel_dielectric(h_dot_in, eta);
    if (rw <= F) {
        s.dir_out = normalize(-in + 2 * dot(in, h) * h); s.eta = 0; s.roughness = roughness;
        return true;
    }
    double h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
    if (h_dot_out_sq <= 0) return false;
    if (h_dot_in < 0) h = -h;
    double h_dot_out = sqrt(h_dot_out_sq);
    s.dir_out = -in / eta + (fabs(h_dot_in) / eta - h_dot_out) * h; s.eta = eta; s.roughness = roughness;
    return true;
}

// ---- DisneyBSDF, src/materials/disney_bsdf.inl ----------------------------------------------------
struct DisneyParams { D3 c_0; double spec_trans, metallic, clearcoat, sheen, eta; };
GD DisneyParams disney_params(const Ctx &c, const GdptMaterial &m, D3 in) {
    DisneyParams p;
    D3 bc = T3(c, m.tex[0]);
    p.spec_trans = T1(c, m.tex[1]); p.metallic = T1(c, m.tex[2]);
    double specular = T1(c, m.tex[4]), specular_tint = T1(c, m.tex[6]);
    p.sheen = T1(c, m.tex[8]); p.clearcoat = T1(c, m.tex[10]);
    double lum = luminance(bc);
    D3 c_tint = (lum > 0) ? bc / lum : splat(1.0);
    p.eta = dot(c.v.gn, in) > 0 ? m.eta : 1 / m.eta;
    D3 K_s = splat(1 - specular_tint) + specular_tint * c_tint;
    double r_0 = sqr(p.eta - 1) / sqr(p.eta + 1);
    p.c_0 = specular * r_0 * (1 - p.metallic) * K_s + p.metallic * bc;
    return p;
}
GD D3 db_eval(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {
    bool inside = dot(c.v.gn, in) <= 0;
    DisneyParams p = disney_params(c, m, in);
    D3 glass = dg_eval(c, m.tex[0], m.tex[5], m.tex[7], p.eta, in, out);   // eta flipped twice (disney_bsdf.inl:29,38)
    double wg = (1 - p.metallic) * p.spec_trans;
    if (inside) return wg * glass;
    double wd = (1 - p.spec_trans) * (1 - p.metallic), wm = (1 - p.spec_trans * (1 - p.metallic));
    double wc = 0.25 * p.clearcoat, ws = (1 - p.metallic) * p.sheen;
    D3 fd = dd_eval(c, m.tex[0], m.tex[5], m.tex[3], in, out);
    D3 fm = dm_eval(c, p.c_0, m.tex[5], m.tex[7], in, out);
    D3 fs = sh_eval(c, m.tex[0], m.tex[9], in, out);
    D3 fc = cc_eval(c, m.tex[11], in, out);
    return wd * fd + wm * fm + wc * fc + wg * glass + ws * fs;
}
GD double db_pdf(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {
    bool inside = dot(c.v.gn, in) <= 0;
    DisneyParams p = disney_params(c, m, in);
    if (inside) return dg_pdf(c, m.tex[5], m.tex[7], p.eta, in, out);
    double wd = (1 - p.spec_trans) * (1 - p.metallic), wm = (1 - p.spec_trans * (1 - p.metallic));
    double wc = 0.25 * p.clearcoat, wg = (1 - p.metallic) * p.spec_trans;
    double net = wd + wm + wc + wg;
    return (wd / net) * cos_pdf(c, in, out) + (wm / net) * dm_pdf(c, m.tex[5], m.tex[7], in, out) +
           (wc / net) * cc_pdf(c, m.tex[11], in, out) + (wg / net) * dg_pdf(c, m.tex[5], m.tex[7], p.eta, in, out);
}
// db_eval and db_pdf in one pass: the parameter block and the glass lobe's terms are formed once (the remaining lobes are
// called as before: their eval and pdf differ in the roughness clamp or the cosine they divide by).
GD void db_eval_pdf(const Ctx &c, const GdptMaterial &m, D3 in, D3 out, D3 &f, double &pdf) {
    bool inside = dot(c.v.gn, in) <= 0;
    DisneyParams p = disney_params(c, m, in);
    D3 glass; double glass_pdf;
    dg_eval_pdf(c, m.tex[0], m.tex[5], m.tex[7], p.eta, in, out, glass, glass_pdf);   // eta flipped twice (disney_bsdf.inl:29,38)
    double wg = (1 - p.metallic) * p.spec_trans;
    if (inside) { f = wg * glass; pdf = glass_pdf; return; }
    double wd = (1 - p.spec_trans) * (1 - p.metallic), wm = (1 - p.spec_trans * (1 - p.metallic));
    double wc = 0.25 * p.clearcoat, ws = (1 - p.metallic) * p.sheen;
    D3 fd = dd_eval(c, m.tex[0], m.tex[5], m.tex[3], in, out);
    D3 fm = dm_eval(c, p.c_0, m.tex[5], m.tex[7], in, out);
    D3 fs = sh_eval(c, m.tex[0], m.tex[9], in, out);
    D3 fc = cc_eval(c, m.tex[11], in, out);
    f = wd * fd + wm * fm + wc * fc + wg * glass + ws * fs;
    double net = wd + wm + wc + wg;
    pdf = (wd / net) * cos_pdf(c, in, out) + (wm / net) * dm_pdf(c, m.tex[5], m.tex[7], in, out) +
          (wc / net) * cc_pdf(c, m.tex[11], in, out) + (wg / net) * glass_pdf;
}
GD bool db_sample(const Ctx &c, const GdptMaterial &m, D3 in, D2 ruv, double rw, BsdfSample &s) {
    double r = ruv.x;   // fixed thresholds; the number is reused unrescaled (disney_bsdf.inl:173-191)
    if (r < 0.25) return dd_sample(c, m.tex[5], in, ruv, s);
    if (r < 0.5) return dm_sample(c, m.tex[5], m.tex[7], in, ruv, s);
    if (r < 0.75) return cc_sample(c, m.tex[11], in, ruv, s);
    DisneyParams p = disney_params(c, m, in);
    return dg_sample(c, m.tex[5], p.eta, in, ruv, rw, s);
}

// ---- RoughPlastic / RoughDielectric: src/materials/roughplastic.inl, roughdielectric.inl ----
GD double gtr2_iso(double n_dot_h, double roughness) {               // GTR2, src/microfacet.h:58-63
    double alpha = roughness * roughness;
    double a2 = alpha * alpha;
    double t = 1 + (a2 - 1) * n_dot_h * n_dot_h;
    return a2 / (kPi * t * t);
}
GD double smith_gtr2_iso(D3 v_local, double roughness) {             // smith_masking_gtr2, src/microfacet.h:72-78
    double alpha = roughness * roughness;
    double a2 = alpha * alpha;
    D3 v2 = v_local * v_local;
    double Lambda = (-1 + sqrt(1 + (v2.x * a2 + v2.y * a2) / v2.z)) / 2;
    return 1 / (1 + Lambda);
}
GD double clamp_rough(double r) { return fmin(fmax(r, 0.01), 1.0); }
GD D3 rp_eval(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {                      // roughplastic.inl:3-43
    if (dot(c.v.gn, in) < 0 || dot(c.v.gn, out) < 0) return splat(0);
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_dot_h = dot(f.n, h), n_dot_in = dot(f.n, in), n_dot_out = dot(f.n, out);
    if (n_dot_out <= 0 || n_dot_h <= 0) return splat(0);
    D3 Kd = T3(c, m.tex[0]), Ks = T3(c, m.tex[1]);
    double roughness = clamp_rough(T1(c, m.tex[2]));
    double F_o = fresnel_dielectric(dot(h, out), m.eta);
    double D = gtr2_iso(n_dot_h, roughness);
    double G = smith_gtr2_iso(to_local(f, in), roughness) * smith_gtr2_iso(to_local(f, out), roughness);
    D3 spec = Ks * (G * F_o * D) / (4 * n_dot_in * n_dot_out);
    double F_i = fresnel_dielectric(dot(h, in), m.eta);
    D3 diff = Kd * (1.0 - F_o) * (1.0 - F_i) / kPi;
    return (spec + diff) * n_dot_out;
}
GD double rp_pdf(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {                   // roughplastic.inl:45-88
    if (dot(c.v.gn, in) < 0 || dot(c.v.gn, out) < 0) return 0;
    Frame f = oriented_frame(c.v, in);
    D3 h = normalize(in + out);
    double n_dot_in = dot(f.n, in), n_dot_out = dot(f.n, out), n_dot_h = dot(f.n, h);
    if (n_dot_out <= 0 || n_dot_h <= 0) return 0;
    double lS = luminance(T3(c, m.tex[1])), lR = luminance(T3(c, m.tex[0]));
    if (lS + lR <= 0) return 0;
    double roughness = clamp_rough(T1(c, m.tex[2]));
    double spec_prob = lS / (lS + lR);
    double diff_prob = 1 - spec_prob;
    double G = smith_gtr2_iso(to_local(f, in), roughness);
    double D = gtr2_iso(n_dot_h, roughness);
    spec_prob *= (G * D) / (4 * n_dot_in);
    diff_prob *= n_dot_out / kPi;
    return spec_prob + diff_prob;
}
GD bool rp_sample(const Ctx &c, const GdptMaterial &m, D3 in, D2 ruv, double rw, BsdfSample &s) {   // :90-137
    if (dot(c.v.gn, in) < 0) return false;
    Frame f = oriented_frame(c.v, in);
    double lS = luminance(T3(c, m.tex[1])), lR = luminance(T3(c, m.tex[0]));
    if (lS + lR <= 0) return false;
    double spec_prob = lS / (lS + lR);
    if (rw < spec_prob) {
        double roughness = clamp_rough(T1(c, m.tex[2]));
        double alpha = roughness * roughness;
        D3 h = to_world(f, sample_visible_normals(to_local(f, in), alpha, alpha, ruv));
        s.dir_out = normalize(-in + 2 * dot(in, h) * h); s.eta = 0; s.roughness = roughness;
    } else {
        s.dir_out = to_world(f, sample_cos_hemisphere(ruv)); s.eta = 0; s.roughness = 1;
    }
    return true;
}
struct RdTerms { bool reflect; Frame f; double eta, roughness, h_dot_in, F, D; D3 h; };
GD RdTerms rd_terms(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {                // roughdielectric.inl:4-37,52-78
    RdTerms t;
    t.reflect = dot(c.v.gn, in) * dot(c.v.gn, out) > 0;
    t.f = oriented_frame_2s(c.v, in);
    t.eta = dot(c.v.gn, in) > 0 ? m.eta : 1 / m.eta;
    t.h = t.reflect ? normalize(in + out) : normalize(in + out * t.eta);
    if (dot(t.h, t.f.n) < 0) t.h = -t.h;
    t.roughness = clamp_rough(T1(c, m.tex[2]));
    t.h_dot_in = dot(t.h, in);
    t.F = fresnel_dielectric(t.h_dot_in, t.eta);
    t.D = gtr2_iso(dot(t.f.n, t.h), t.roughness);
    return t;
}
GD D3 rd_eval(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {                      // roughdielectric.inl:3-49
    RdTerms t = rd_terms(c, m, in, out);
    D3 Ks = T3(c, m.tex[0]), Kt = T3(c, m.tex[1]);
    double G = smith_gtr2_iso(to_local(t.f, in), t.roughness) * smith_gtr2_iso(to_local(t.f, out), t.roughness);
    if (t.reflect) return Ks * (t.F * t.D * G) / (4 * fabs(dot(t.f.n, in)));
    double eta_factor = 1 / (t.eta * t.eta);                  // TransportDirection::TO_LIGHT, the default of eval()
    double h_dot_out = dot(t.h, out);
    double sd = t.h_dot_in + t.eta * h_dot_out;
    return Kt * (eta_factor * (1 - t.F) * t.D * G * t.eta * t.eta * fabs(h_dot_out * t.h_dot_in)) / (fabs(dot(t.f.n, in)) * sd * sd);
}
GD double rd_pdf(const Ctx &c, const GdptMaterial &m, D3 in, D3 out) {                   // roughdielectric.inl:51-93
    RdTerms t = rd_terms(c, m, in, out);
    double G_in = smith_gtr2_iso(to_local(t.f, in), t.roughness);
    if (t.reflect) return (t.F * t.D * G_in) / (4 * fabs(dot(t.f.n, in)));
    double h_dot_out = dot(t.h, out);
    double sd = t.h_dot_in + t.eta * h_dot_out;
    double dh_dout = t.eta * t.eta * h_dot_out / (sd * sd);
    return (1 - t.F) * t.D * G_in * fabs(dh_dout * t.h_dot_in / dot(t.f.n, in));
}
GD bool rd_sample(const Ctx &c, const GdptMaterial &m, D3 in, D2 ruv, double rw, BsdfSample &s) {   // :95-139
    double eta = dot(c.v.gn, in) > 0 ? m.eta : 1 / m.eta;
    Frame f = oriented_frame_2s(c.v, in);
    double roughness = clamp_rough(T1(c, m.tex[2]));
    double alpha = roughness * roughness;
    D3 h = to_world(f, sample_visible_normals(to_local(f, in), alpha, alpha, ruv));
    if (dot(h, f.n) < 0) h = -h;
    double h_dot_in = dot(h, in);
    double F = fresnel_dielectric(h_dot_in, eta);
    if (rw <= F) {
        s.dir_out = normalize(-in + 2 * dot(in, h) * h); s.eta = 0; s.roughness = roughness;
        return true;
    }
    double h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
    if (h_dot_out_sq <= 0) return false;
    if (h_dot_in < 0) h = -h;
    double h_dot_out = sqrt(h_dot_out_sq);
    s.dir_out = -in / eta + (fabs(h_dot_in) / eta - h_dot_out) * h; s.eta = eta; s.roughness = roughness;
    return true;
}

// ---- dispatch (std::visit in the reference, src/material.cpp:90-119) -------------------------------
// ROUGH = false drops the RoughPlastic / RoughDielectric cases: inlined into the switch they cost scenes that never take
// them 20 % (Disney test scenes in the GradPath lane machine), so that kernel is built without them and scenes that do
// use them go to the kernels built with ROUGH = true.
// TWOSIDED = false likewise drops DisneyGlass / DisneyBSDF (the two heaviest lobes): the one-sided GradPath lane
// machine never meets them.
// MASK: bit t set = material type t may occur (decided from the scene at upload, gdpt_scene_upload): a kernel built for
// {Lambertian, DisneyGlass} does not carry DisneyBSDF's five inlined lobes through its register allocation.
constexpr unsigned kAllMaterials = 0x1FFu;
template <unsigned MASK, int T> GD constexpr bool mat_on() { return (MASK >> T) & 1u; }
template <bool ROUGH = true, bool TWOSIDED = true, unsigned MASK = kAllMaterials>
GD D3 bsdf_eval(const DevSceneView &sv, const GdptMaterial &m, D3 in, D3 out, const Vertex &v) {
    Ctx c{sv, v};
    switch (m.type) {
        case GDPT_MAT_ROUGHPLASTIC: if (ROUGH) return rp_eval(c, m, in, out); else return splat(0);
        case GDPT_MAT_ROUGHDIELECTRIC: if (ROUGH) return rd_eval(c, m, in, out); else return splat(0);
        case GDPT_MAT_LAMBERTIAN: return lambert_eval(c, m.tex[0], in, out);
        case GDPT_MAT_DISNEY_DIFFUSE: if (mat_on<MASK, GDPT_MAT_DISNEY_DIFFUSE>()) return dd_eval(c, m.tex[0], m.tex[1], m.tex[2], in, out); else return splat(0);
        case GDPT_MAT_DISNEY_METAL: if (mat_on<MASK, GDPT_MAT_DISNEY_METAL>()) return dm_eval(c, T3(c, m.tex[0]), m.tex[1], m.tex[2], in, out); else return splat(0);
        case GDPT_MAT_DISNEY_GLASS: if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_GLASS>()) return dg_eval(c, m.tex[0], m.tex[1], m.tex[2], m.eta, in, out); else return splat(0);
        case GDPT_MAT_DISNEY_CLEARCOAT: if (mat_on<MASK, GDPT_MAT_DISNEY_CLEARCOAT>()) return cc_eval(c, m.tex[0], in, out); else return splat(0);
        case GDPT_MAT_DISNEY_SHEEN: if (mat_on<MASK, GDPT_MAT_DISNEY_SHEEN>()) return sh_eval(c, m.tex[0], m.tex[1], in, out); else return splat(0);
        case GDPT_MAT_DISNEY_BSDF: if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_BSDF>()) return db_eval(c, m, in, out); else return splat(0);
        default: return splat(0);
    }
}
template <bool ROUGH = true, bool TWOSIDED = true, unsigned MASK = kAllMaterials>
GD double bsdf_pdf(const DevSceneView &sv, const GdptMaterial &m, D3 in, D3 out, const Vertex &v) {
    Ctx c{sv, v};
    switch (m.type) {
        case GDPT_MAT_ROUGHPLASTIC: if (ROUGH) return rp_pdf(c, m, in, out); else return 0;
        case GDPT_MAT_ROUGHDIELECTRIC: if (ROUGH) return rd_pdf(c, m, in, out); else return 0;
        case GDPT_MAT_LAMBERTIAN: case GDPT_MAT_DISNEY_DIFFUSE: case GDPT_MAT_DISNEY_SHEEN: return cos_pdf(c, in, out);
        case GDPT_MAT_DISNEY_METAL: if (mat_on<MASK, GDPT_MAT_DISNEY_METAL>()) return dm_pdf(c, m.tex[1], m.tex[2], in, out); else return 0;
        case GDPT_MAT_DISNEY_GLASS: if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_GLASS>()) return dg_pdf(c, m.tex[1], m.tex[2], m.eta, in, out); else return 0;
        case GDPT_MAT_DISNEY_CLEARCOAT: if (mat_on<MASK, GDPT_MAT_DISNEY_CLEARCOAT>()) return cc_pdf(c, m.tex[0], in, out); else return 0;
        case GDPT_MAT_DISNEY_BSDF: if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_BSDF>()) return db_pdf(c, m, in, out); else return 0;
        default: return 0;
    }
}
// eval and pdf of one direction pair (what every bounce needs): the two heavy two-sided lobes share their terms
template <bool ROUGH = true, bool TWOSIDED = true, unsigned MASK = kAllMaterials>
GD void bsdf_eval_pdf(const DevSceneView &sv, const GdptMaterial &m, D3 in, D3 out, const Vertex &v, D3 &f, double &pdf) {
    if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_GLASS>() && m.type == GDPT_MAT_DISNEY_GLASS) { Ctx c{sv, v}; dg_eval_pdf(c, m.tex[0], m.tex[1], m.tex[2], m.eta, in, out, f, pdf); return; }
    if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_BSDF>() && m.type == GDPT_MAT_DISNEY_BSDF) { Ctx c{sv, v}; db_eval_pdf(c, m, in, out, f, pdf); return; }
    f = bsdf_eval<ROUGH, TWOSIDED, MASK & ~((1u << GDPT_MAT_DISNEY_GLASS) | (1u << GDPT_MAT_DISNEY_BSDF))>(sv, m, in, out, v);
    pdf = bsdf_pdf<ROUGH, TWOSIDED, MASK & ~((1u << GDPT_MAT_DISNEY_GLASS) | (1u << GDPT_MAT_DISNEY_BSDF))>(sv, m, in, out, v);
}
template <bool ROUGH = true, bool TWOSIDED = true, unsigned MASK = kAllMaterials>
GD bool bsdf_sample(const DevSceneView &sv, const GdptMaterial &m, D3 in, const Vertex &v, D2 ruv, double rw, BsdfSample &s) {
    Ctx c{sv, v};
    switch (m.type) {
        case GDPT_MAT_ROUGHPLASTIC: if (ROUGH) return rp_sample(c, m, in, ruv, rw, s); else return false;
        case GDPT_MAT_ROUGHDIELECTRIC: if (ROUGH) return rd_sample(c, m, in, ruv, rw, s); else return false;
        case GDPT_MAT_LAMBERTIAN: case GDPT_MAT_DISNEY_SHEEN: return cos_sample(c, in, ruv, 1.0, s);
        case GDPT_MAT_DISNEY_DIFFUSE: if (mat_on<MASK, GDPT_MAT_DISNEY_DIFFUSE>()) return dd_sample(c, m.tex[1], in, ruv, s); else return false;
        case GDPT_MAT_DISNEY_METAL: if (mat_on<MASK, GDPT_MAT_DISNEY_METAL>()) return dm_sample(c, m.tex[1], m.tex[2], in, ruv, s); else return false;
        case GDPT_MAT_DISNEY_GLASS: if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_GLASS>()) return dg_sample(c, m.tex[1], m.eta, in, ruv, rw, s); else return false;
        case GDPT_MAT_DISNEY_CLEARCOAT: if (mat_on<MASK, GDPT_MAT_DISNEY_CLEARCOAT>()) return cc_sample(c, m.tex[0], in, ruv, s); else return false;
        case GDPT_MAT_DISNEY_BSDF: if (TWOSIDED && mat_on<MASK, GDPT_MAT_DISNEY_BSDF>()) return db_sample(c, m, in, ruv, rw, s); else return false;
        default: return false;
    }
}

} // namespace gd
