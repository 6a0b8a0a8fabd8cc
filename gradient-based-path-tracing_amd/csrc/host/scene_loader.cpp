// scene_loader.cpp — Mitsuba-0.x XML subset -> GdptSceneDesc.
// Host-side mirror of the reference's scene ingest for the GradPath hot path:
//   parse_scene            src/parsers/parse_scene.cpp:1405-1630
//   parse_sensor/film      src/parsers/parse_scene.cpp:604-860
//   parse_bsdf             src/parsers/parse_scene.cpp:935-1196
//   parse_shape            src/parsers/parse_scene.cpp:1198-1403
//   parse_obj              src/parsers/parse_obj.cpp
//   load_serialized        src/parsers/load_serialized.cpp
//   compute_normal         src/parsers/shape_utils.h
//   spectrum -> RGB        src/spectrum.h:72-118
// Quirks kept on purpose (they change pixel values): every XML float goes through
// std::stof (fp32) and is widened (parse_scene.cpp:90-96); a one-entry <spectrum>
// reflectance becomes (1,1,1) whatever its value (parse_scene.cpp:300-301,393-394);
// sphere shapes ignore toWorld (parse_scene.cpp:1338-1352).
#include "scene_loader.h"
#include "image_io.h"
#include "xml_lite.h"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace fs = std::filesystem;

namespace gdpt {

namespace {

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error(msg); }

using DefaultMap = std::map<std::string, std::string>;

// std::sregex_token_iterator(..., "(,| )+", -1) semantics without <regex>:
// a leading delimiter run yields an empty first token, trailing runs yield nothing.
std::vector<std::string> split_list(const std::string &s, const char *delims) {
    std::vector<std::string> out;
    size_t i = 0, n = s.size();
    auto is_delim = [&](char c) { return std::strchr(delims, c) != nullptr && c != '\0'; };
    if (n == 0) return out;
    size_t b = 0;
    while (i < n) {
        if (is_delim(s[i])) {
            out.push_back(s.substr(b, i - b));
            while (i < n && is_delim(s[i])) i++;
            b = i;
        } else {
            i++;
        }
    }
    if (b < n) out.push_back(s.substr(b));
    return out;
}

// split on single '/' (std::regex("/")): empty fields kept, trailing empty dropped
std::vector<std::string> split_slash(const std::string &s) {
    std::vector<std::string> out;
    size_t b = 0;
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '/') { out.push_back(s.substr(b, i - b)); b = i + 1; }
    }
    if (b < s.size()) out.push_back(s.substr(b));
    return out;
}

const std::string &subst(const std::string &value, const DefaultMap &dm) {
    if (!value.empty() && value[0] == '$') {
        auto it = dm.find(value.substr(1));
        if (it == dm.end()) fail("Reference default variable " + value + " not found.");
        return it->second;
    }
    return value;
}

double to_float(const std::string &s) {
    try { return (double)std::stof(s); }
    catch (const std::exception &) { fail("stof: cannot parse float from \"" + s + "\""); }
}
int to_int(const std::string &s) {
    try { return std::stoi(s); }
    catch (const std::exception &) { fail("stoi: cannot parse integer from \"" + s + "\""); }
}
double parse_float(const std::string &v, const DefaultMap &dm) { return to_float(subst(v, dm)); }
int parse_integer(const std::string &v, const DefaultMap &dm) { return to_int(subst(v, dm)); }
std::string parse_string(const std::string &v, const DefaultMap &dm) { return subst(v, dm); }
bool parse_boolean(const std::string &v, const DefaultMap &dm) {
    const std::string &s = subst(v, dm);
    if (s == "true") return true;
    if (s == "false") return false;
    fail("parse_boolean failed");
}
V3 parse_vector3(const std::string &v, const DefaultMap &dm) {
    std::vector<std::string> l = split_list(subst(v, dm), ", ");
    if (l.size() == 1) { double f = to_float(l[0]); return {f, f, f}; }
    if (l.size() == 3) return {to_float(l[0]), to_float(l[1]), to_float(l[2])};
    fail("parse_vector3 failed");
}
V3 parse_srgb(const std::string &v, const DefaultMap &dm) {
    const std::string &s = subst(v, dm);
    if (s.size() == 7 && s[0] == '#') {
        char *end = nullptr;
        long enc = std::strtol(s.c_str() + 1, &end, 16);
        if (*end != '\0') fail("Invalid SRGB value: " + s);
        return {(double)(((enc & 0xFF0000) >> 16) / 255.0f), (double)(((enc & 0x00FF00) >> 8) / 255.0f),
                (double)((enc & 0x0000FF) / 255.0f)};
    }
    fail("Unknown SRGB format: " + s);
}
std::vector<std::pair<double, double>> parse_spectrum(const std::string &v, const DefaultMap &dm) {
    std::vector<std::string> l = split_list(subst(v, dm), ", ");
    std::vector<std::pair<double, double>> s;
    if (l.size() == 1 && l[0].find(':') == std::string::npos) {
        s.emplace_back(-1.0, to_float(l[0]));
    } else {
        for (auto &tok : l) {
            size_t c = tok.find(':');
            if (c == std::string::npos || c + 1 > tok.size()) fail("parse_spectrum failed");
            std::string a = tok.substr(0, c), b = tok.substr(c + 1);
            size_t c2 = b.find(':');
            if (c2 != std::string::npos) b = b.substr(0, c2);
            if (b.empty()) fail("parse_spectrum failed");
            s.emplace_back(to_float(a), to_float(b));
        }
    }
    return s;
}
M4 parse_matrix4x4(const std::string &v, const DefaultMap &dm) {
    std::vector<std::string> l = split_list(subst(v, dm), ", ");
    if (l.size() != 16) fail("parse_matrix4x4 failed");
    M4 m{};
    int k = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = to_float(l[k++]);
    return m;
}

std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}

M4 parse_transform(const XmlNode &node, const DefaultMap &dm) {
    M4 t = M4::identity();
    for (auto &cp : node.children) {
        const XmlNode &c = *cp;
        std::string name = lower(c.name);
        if (name == "scale" || name == "translate") {
            double d = (name == "scale") ? 1.0 : 0.0;
            double x = d, y = d, z = d;
            if (c.has_attr("x")) x = parse_float(c.attr("x"), dm);
            if (c.has_attr("y")) y = parse_float(c.attr("y"), dm);
            if (c.has_attr("z")) z = parse_float(c.attr("z"), dm);
            if (c.has_attr("value")) { V3 v = parse_vector3(c.attr("value"), dm); x = v.x; y = v.y; z = v.z; }
            t = (name == "scale" ? scale(V3{x, y, z}) : translate(V3{x, y, z})) * t;
        } else if (name == "rotate") {
            double x = 0, y = 0, z = 0, angle = 0;
            if (c.has_attr("x")) x = parse_float(c.attr("x"), dm);
            if (c.has_attr("y")) y = parse_float(c.attr("y"), dm);
            if (c.has_attr("z")) z = parse_float(c.attr("z"), dm);
            if (c.has_attr("angle")) angle = parse_float(c.attr("angle"), dm);
            t = rotate(angle, V3{x, y, z}) * t;
        } else if (name == "lookat") {
            V3 pos = parse_vector3(c.attr("origin"), dm);
            V3 target = parse_vector3(c.attr("target"), dm);
            V3 up = parse_vector3(c.attr("up"), dm);
            t = look_at(pos, target, up) * t;
        } else if (name == "matrix") {
            t = parse_matrix4x4(c.attr("value"), dm) * t;
        }
    }
    return t;
}

GdptTexture const_spectrum_tex(const V3 &c) {
    GdptTexture t{};
    t.type = GDPT_TEX_CONSTANT; t.image_id = -1;
    t.v0[0] = c.x; t.v0[1] = c.y; t.v0[2] = c.z;
    t.uscale = t.vscale = 1;
    return t;
}
GdptTexture const_float_tex(double f) { return const_spectrum_tex(V3{f, f, f}); }

struct ParsedTexture {
    bool bitmap = false;
    std::string filename;
    V3 color0{0.4, 0.4, 0.4}, color1{0.2, 0.2, 0.2};
    double uscale = 1, vscale = 1, uoffset = 0, voffset = 0;
};

V3 parse_color(const XmlNode &node, const DefaultMap &dm) {
    const std::string &type = node.name;
    if (type == "spectrum") {
        auto spec = parse_spectrum(node.attr("value"), dm);
        if (spec.size() > 1) return spectrum_to_rgb(spec);
        if (spec.size() == 1) return {1, 1, 1};
        return {0, 0, 0};
    } else if (type == "rgb") {
        return parse_vector3(node.attr("value"), dm);
    } else if (type == "srgb") {
        V3 s = parse_srgb(node.attr("value"), dm), rgb = s;
        for (int i = 0; i < 3; i++)
            rgb[i] = s[i] <= 0.04045 ? s[i] / 12.92 : std::pow((s[i] + 0.055) / 1.055, 2.4);
        return rgb;
    } else if (type == "float") {
        double f = parse_float(node.attr("value"), dm);
        return {f, f, f};
    }
    fail("Unknown color type:" + type);
}

// XYZ(0.9505,1,1.0888) white point for single-valued emitter spectra (parse_scene.cpp:500-527)
V3 xyz_to_rgb(const V3 &xyz) {
    return {3.240479 * xyz.x - 1.537150 * xyz.y - 0.498535 * xyz.z,
            -0.969256 * xyz.x + 1.875991 * xyz.y + 0.041556 * xyz.z,
            0.055648 * xyz.x - 0.204043 * xyz.y + 1.057311 * xyz.z};
}
V3 parse_intensity(const XmlNode &node, const DefaultMap &dm) {
    const std::string &t = node.name;
    if (t == "spectrum") {
        auto spec = parse_spectrum(node.attr("value"), dm);
        if (spec.size() == 1) return xyz_to_rgb(V3{0.9505, 1.0, 1.0888} * spec[0].second);
        return spectrum_to_rgb(spec);
    } else if (t == "rgb") {
        return parse_vector3(node.attr("value"), dm);
    } else if (t == "srgb") {
        V3 s = parse_srgb(node.attr("value"), dm), rgb = s;
        for (int i = 0; i < 3; i++)
            rgb[i] = s[i] <= 0.04045 ? s[i] / 12.92 : std::pow((s[i] + 0.055) / 1.055, 2.4);
        return rgb;
    }
    return {1, 1, 1};
}

ParsedTexture parse_texture(const XmlNode &node, const DefaultMap &dm) {
    std::string type = node.attr("type");
    ParsedTexture t;
    if (type != "bitmap" && type != "checkerboard") fail("Unknown texture type: " + type);
    t.bitmap = (type == "bitmap");
    for (auto &cp : node.children) {
        const XmlNode &c = *cp;
        std::string name = c.attr("name");
        if (name == "filename" && t.bitmap) t.filename = parse_string(c.attr("value"), dm);
        else if (name == "color0" && !t.bitmap) t.color0 = parse_color(c, dm);
        else if (name == "color1" && !t.bitmap) t.color1 = parse_color(c, dm);
        else if (name == "uvscale") t.uscale = t.vscale = parse_float(c.attr("value"), dm);
        else if (name == "uscale") t.uscale = parse_float(c.attr("value"), dm);
        else if (name == "vscale") t.vscale = parse_float(c.attr("value"), dm);
        else if (name == "uoffset") t.uoffset = parse_float(c.attr("value"), dm);
        else if (name == "voffset") t.voffset = parse_float(c.attr("value"), dm);
    }
    return t;
}

struct Builder {
    HostScene &hs;
    DefaultMap dm;
    int film_w = 0, film_h = 0;       // > 0: replaces the <film> extent (benchmark configurations quote their own)
    std::map<std::string, int> material_map;
    std::map<std::string, ParsedTexture> texture_map;
    std::map<std::string, int> image3_map, image1_map; // TexturePool::image3s_map / image1s_map (src/texture.h:10-16)
    fs::path scene_dir;

    explicit Builder(HostScene &h) : hs(h) {}

    int insert_image(const std::string &key, const std::string &filename, int channels) {
        auto &m = (channels == 3) ? image3_map : image1_map;
        auto it = m.find(key);
        if (it != m.end()) return it->second;
        int w = 0, h = 0;
        std::vector<double> texels;
        load_texture_file((scene_dir / filename).string(), channels, &w, &h, &texels);
        int id = (int)hs.images.size();
        hs.image_data.push_back(std::move(texels));
        GdptImage img{};
        img.width = w; img.height = h; img.channels = channels;
        img.texels = hs.image_data.back().data();
        hs.images.push_back(img);
        m[key] = id;
        return id;
    }

    GdptTexture tex_from_parsed(const ParsedTexture &t, const std::string &key, bool spectrum) {
        GdptTexture g{};
        g.uscale = t.uscale; g.vscale = t.vscale; g.uoffset = t.uoffset; g.voffset = t.voffset;
        g.image_id = -1;
        if (t.bitmap) {
            g.type = GDPT_TEX_IMAGE;
            g.image_id = insert_image(key, t.filename, spectrum ? 3 : 1);
        } else {
            g.type = GDPT_TEX_CHECKERBOARD;
            if (spectrum) {
                for (int i = 0; i < 3; i++) { g.v0[i] = t.color0[i]; g.v1[i] = t.color1[i]; }
            } else { // avg(), src/spectrum.h:37-39
                double a0 = (t.color0.x + t.color0.y + t.color0.z) / 3, a1 = (t.color1.x + t.color1.y + t.color1.z) / 3;
                for (int i = 0; i < 3; i++) { g.v0[i] = a0; g.v1[i] = a1; }
            }
        }
        return g;
    }

    int inline_counter = 0;

    GdptTexture parse_spectrum_texture(const XmlNode &node) {
        const std::string &type = node.name;
        if (type == "spectrum" || type == "rgb" || type == "srgb") return const_spectrum_tex(parse_color(node, dm));
        if (type == "ref") {
            std::string id = node.attr("id");
            auto it = texture_map.find(id);
            if (it == texture_map.end()) fail("Texture not found. ID = " + id);
            return tex_from_parsed(it->second, id, true);
        }
        if (type == "texture") {
            ParsedTexture t = parse_texture(node, dm);
            return tex_from_parsed(t, "$inline_spectrum_texture" + std::to_string(inline_counter++), true);
        }
        fail("Unknown spectrum texture type:" + type);
    }
    GdptTexture parse_float_texture(const XmlNode &node) {
        const std::string &type = node.name;
        if (type == "ref") {
            std::string id = node.attr("id");
            auto it = texture_map.find(id);
            if (it == texture_map.end()) fail("Texture not found. ID = " + id);
            return tex_from_parsed(it->second, id, false);
        }
        if (type == "float") return const_float_tex(parse_float(node.attr("value"), dm));
        if (type == "texture") {
            ParsedTexture t = parse_texture(node, dm);
            return tex_from_parsed(t, "$inline_float_texture" + std::to_string(inline_counter++), false);
        }
        fail("Unknown float texture type:" + type);
    }
    // microfacet `alpha` -> roughness = sqrt(alpha) (parse_scene.cpp:862-933); constant and checkerboard only
    GdptTexture alpha_to_roughness(const XmlNode &node) {
        const std::string &type = node.name;
        if (type == "float") return const_float_tex(std::sqrt(parse_float(node.attr("value"), dm)));
        ParsedTexture t;
        if (type == "ref") {
            auto it = texture_map.find(node.attr("id"));
            if (it == texture_map.end()) fail("Texture not found. ID = " + node.attr("id"));
            t = it->second;
        } else if (type == "texture") {
            t = parse_texture(node, dm);
        } else {
            fail("Unknown float texture type:" + type);
        }
        if (t.bitmap) fail("alpha bitmap textures are outside the GradPath hot-path subset");
        GdptTexture g = tex_from_parsed(t, "", false);
        for (int i = 0; i < 3; i++) { g.v0[i] = std::sqrt(g.v0[i]); g.v1[i] = std::sqrt(g.v1[i]); }
        return g;
    }

    // returns (id, material); `ok=false` mirrors the reference's ("", Material{}) fall-through
    std::pair<std::string, GdptMaterial> parse_bsdf(const XmlNode &node, const std::string &parent_id = "") {
        std::string type = node.attr("type");
        std::string id = parent_id;
        if (node.has_attr("id")) id = node.attr("id");
        GdptMaterial m{};
        m.eta = 1.5;
        for (auto &t : m.tex) t = const_float_tex(0);
        auto each = [&](auto fn) { for (auto &cp : node.children) fn(*cp, cp->attr("name")); };
        if (type == "twosided") {
            for (auto &cp : node.children)
                if (cp->name == "bsdf") return parse_bsdf(*cp, id);
            m.type = GDPT_MAT_LAMBERTIAN; // reference returns a default-constructed variant = Lambertian
            return {"", m};
        } else if (type == "diffuse") {
            m.type = GDPT_MAT_LAMBERTIAN;
            m.tex[0] = const_spectrum_tex({0.5, 0.5, 0.5});
            each([&](const XmlNode &c, const std::string &n) { if (n == "reflectance") m.tex[0] = parse_spectrum_texture(c); });
        } else if (type == "roughplastic" || type == "plastic") {
            m.type = GDPT_MAT_ROUGHPLASTIC;
            m.tex[0] = const_spectrum_tex({0.5, 0.5, 0.5});
            m.tex[1] = const_spectrum_tex({1, 1, 1});
            m.tex[2] = const_float_tex(type == "plastic" ? 0.01 : 0.1);
            double int_ior = 1.49, ext_ior = 1.000277;
            each([&](const XmlNode &c, const std::string &n) {
                if (n == "diffuseReflectance" || n == "diffuse_reflectance") m.tex[0] = parse_spectrum_texture(c);
                else if (n == "specularReflectance" || n == "specular_reflectance") m.tex[1] = parse_spectrum_texture(c);
                else if (n == "alpha") m.tex[2] = alpha_to_roughness(c);
                else if (n == "roughness") m.tex[2] = parse_float_texture(c);
                else if (n == "intIOR" || n == "int_ior") int_ior = parse_float(c.attr("value"), dm);
                else if (n == "extIOR" || n == "ext_ior") ext_ior = parse_float(c.attr("value"), dm);
            });
            m.eta = int_ior / ext_ior;
        } else if (type == "roughdielectric" || type == "dielectric") {
            m.type = GDPT_MAT_ROUGHDIELECTRIC;
            m.tex[0] = const_spectrum_tex({1, 1, 1});
            m.tex[1] = const_spectrum_tex({1, 1, 1});
            m.tex[2] = const_float_tex(type == "dielectric" ? 0.01 : 0.1);
            double int_ior = 1.5046, ext_ior = 1.000277;
            each([&](const XmlNode &c, const std::string &n) {
                if (n == "specularReflectance" || n == "specular_reflectance") m.tex[0] = parse_spectrum_texture(c);
                else if (n == "specularTransmittance" || n == "specular_transmittance") m.tex[1] = parse_spectrum_texture(c);
                else if (n == "alpha") m.tex[2] = alpha_to_roughness(c);
                else if (n == "roughness") m.tex[2] = parse_float_texture(c);
                else if (n == "intIOR" || n == "int_ior") int_ior = parse_float(c.attr("value"), dm);
                else if (n == "extIOR" || n == "ext_ior") ext_ior = parse_float(c.attr("value"), dm);
            });
            m.eta = int_ior / ext_ior;
        } else if (type == "disneydiffuse") {
            m.type = GDPT_MAT_DISNEY_DIFFUSE;
            m.tex[0] = const_spectrum_tex({0.5, 0.5, 0.5});
            m.tex[1] = const_float_tex(0.5);
            m.tex[2] = const_float_tex(0);
            each([&](const XmlNode &c, const std::string &n) {
                if (n == "baseColor" || n == "base_color") m.tex[0] = parse_spectrum_texture(c);
                else if (n == "roughness") m.tex[1] = parse_float_texture(c);
                else if (n == "subsurface") m.tex[2] = parse_float_texture(c);
            });
        } else if (type == "disneymetal" || type == "disneyglass") {
            m.type = (type == "disneymetal") ? GDPT_MAT_DISNEY_METAL : GDPT_MAT_DISNEY_GLASS;
            m.tex[0] = const_spectrum_tex({0.5, 0.5, 0.5});
            m.tex[1] = const_float_tex(0.5);
            m.tex[2] = const_float_tex(0);
            each([&](const XmlNode &c, const std::string &n) {
                if (n == "baseColor" || n == "base_color") m.tex[0] = parse_spectrum_texture(c);
                else if (n == "roughness") m.tex[1] = parse_float_texture(c);
                else if (n == "anisotropic") m.tex[2] = parse_float_texture(c);
                else if (n == "eta" && type == "disneyglass") m.eta = parse_float(c.attr("value"), dm);
            });
        } else if (type == "disneyclearcoat") {
            m.type = GDPT_MAT_DISNEY_CLEARCOAT;
            m.tex[0] = const_float_tex(1.0);
            each([&](const XmlNode &c, const std::string &n) { if (n == "clearcoatGloss") m.tex[0] = parse_float_texture(c); });
        } else if (type == "disneysheen") {
            m.type = GDPT_MAT_DISNEY_SHEEN;
            m.tex[0] = const_spectrum_tex({0.5, 0.5, 0.5});
            m.tex[1] = const_float_tex(0.5);
            each([&](const XmlNode &c, const std::string &n) {
                if (n == "baseColor" || n == "base_color") m.tex[0] = parse_spectrum_texture(c);
                else if (n == "sheenTint" || n == "sheen_tint") m.tex[1] = parse_float_texture(c);
            });
        } else if (type == "disneybsdf" || type == "principled") {
            m.type = GDPT_MAT_DISNEY_BSDF;
            const double defaults[12] = {0.5, 0, 0, 0, 0.5, 0.5, 0, 0, 0, 0.5, 0, 1};
            for (int i = 0; i < 12; i++) m.tex[i] = const_float_tex(defaults[i]);
            each([&](const XmlNode &c, const std::string &n) {
                if (n == "baseColor" || n == "base_color") m.tex[0] = parse_spectrum_texture(c);
                else if (n == "specularTransmission" || n == "specular_transmission" || n == "specTrans" || n == "spec_trans") m.tex[1] = parse_float_texture(c);
                else if (n == "metallic") m.tex[2] = parse_float_texture(c);
                else if (n == "subsurface") m.tex[3] = parse_float_texture(c);
                else if (n == "specular") m.tex[4] = parse_float_texture(c);
                else if (n == "roughness") m.tex[5] = parse_float_texture(c);
                else if (n == "specularTint" || n == "specular_tint" || n == "specTint" || n == "spec_tint") m.tex[6] = parse_float_texture(c);
                else if (n == "anisotropic") m.tex[7] = parse_float_texture(c);
                else if (n == "sheen") m.tex[8] = parse_float_texture(c);
                else if (n == "sheenTint" || n == "sheen_tint") m.tex[9] = parse_float_texture(c);
                else if (n == "clearcoat") m.tex[10] = parse_float_texture(c);
                else if (n == "clearcoatGloss" || n == "clearcoat_gloss") m.tex[11] = parse_float_texture(c);
                else if (n == "eta") m.eta = parse_float(c.attr("value"), dm);
            });
        } else if (type == "null") {
            m.type = GDPT_MAT_LAMBERTIAN;
            m.tex[0] = const_spectrum_tex({0, 0, 0});
        } else {
            fail("Unknown BSDF: " + type);
        }
        return {id, m};
    }

    void parse_shape(const XmlNode &node) {
        int material_id = -1;
        for (auto &cp : node.children) {
            const XmlNode &c = *cp;
            if (c.name == "ref") {
                std::string name_value = c.attr("name");
                if (!c.has_attr("id")) fail("Material/medium reference id not specified.");
                if (name_value == "interior" || name_value == "exterior") continue; // media: outside GradPath
                auto it = material_map.find(c.attr("id"));
                if (it == material_map.end()) fail("Material reference " + c.attr("id") + " not found.");
                material_id = it->second;
            } else if (c.name == "bsdf") {
                auto [mname, m] = parse_bsdf(c);
                if (!mname.empty()) material_map[mname] = (int)hs.materials.size();
                material_id = (int)hs.materials.size();
                hs.materials.push_back(m);
            }
        }

        GdptShape shape{};
        shape.material_id = material_id;
        shape.area_light_id = -1;
        std::string type = node.attr("type");
        if (type == "obj" || type == "serialized") {
            std::string filename;
            int shape_index = 0;
            M4 to_world = M4::identity();
            bool face_normals = false;
            for (auto &cp : node.children) {
                const XmlNode &c = *cp;
                std::string name = c.attr("name");
                if (name == "filename") filename = parse_string(c.attr("value"), dm);
                else if (name == "toWorld" || name == "to_world") { if (c.name == "transform") to_world = parse_transform(c, dm); }
                else if ((name == "shapeIndex" || name == "shape_index") && type == "serialized") shape_index = parse_integer(c.attr("value"), dm);
                else if (name == "faceNormals" || name == "face_normals") face_normals = parse_boolean(c.attr("value"), dm);
            }
            std::string full = (scene_dir / filename).string();
            HostMesh mesh = (type == "obj") ? load_obj(full, to_world) : load_serialized(full, shape_index, to_world);
            if (face_normals) mesh.normals.clear();
            else if (mesh.normals.empty()) mesh.normals = compute_vertex_normals(mesh.positions, mesh.indices);
            add_mesh(shape, std::move(mesh));
        } else if (type == "sphere") {
            V3 center{0, 0, 0};
            double radius = 1;
            for (auto &cp : node.children) {
                const XmlNode &c = *cp;
                std::string name = c.attr("name");
                if (name == "center") center = V3{parse_float(c.attr("x"), dm), parse_float(c.attr("y"), dm), parse_float(c.attr("z"), dm)};
                else if (name == "radius") radius = parse_float(c.attr("value"), dm);
            }
            shape.type = GDPT_SHAPE_SPHERE;
            shape.center[0] = center.x; shape.center[1] = center.y; shape.center[2] = center.z;
            shape.radius = radius;
        } else if (type == "rectangle") {
            M4 to_world = M4::identity();
            bool flip = false;
            for (auto &cp : node.children) {
                const XmlNode &c = *cp;
                std::string name = c.attr("name");
                if (name == "toWorld" || name == "to_world") { if (c.name == "transform") to_world = parse_transform(c, dm); }
                else if (name == "flipNormals" || name == "flip_normals") flip = parse_boolean(c.attr("value"), dm);
            }
            HostMesh mesh;
            const V3 P[4] = {{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 1, 0}};
            const double UV[8] = {0, 0, 1, 0, 1, 1, 0, 1};
            M4 inv = inverse(to_world);
            for (int i = 0; i < 4; i++) {
                V3 p = xform_point(to_world, P[i]);
                V3 n = xform_normal(inv, flip ? V3{0, 0, -1} : V3{0, 0, 1});
                mesh.positions.insert(mesh.positions.end(), {p.x, p.y, p.z});
                mesh.normals.insert(mesh.normals.end(), {n.x, n.y, n.z});
            }
            mesh.uvs.assign(UV, UV + 8);
            mesh.indices = {0, 1, 2, 0, 2, 3};
            add_mesh(shape, std::move(mesh));
        } else {
            fail("Unknown shape:" + type);
        }

        for (auto &cp : node.children) {
            const XmlNode &c = *cp;
            if (c.name == "emitter") {
                V3 radiance{1, 1, 1};
                for (auto &gp : c.children)
                    if (gp->attr("name") == "radiance") radiance = parse_intensity(*gp, dm);
                shape.area_light_id = (int)hs.lights.size();
                GdptLight l{};
                l.shape_id = (int)hs.shapes.size();
                l.intensity[0] = radiance.x; l.intensity[1] = radiance.y; l.intensity[2] = radiance.z;
                hs.lights.push_back(l);
            }
        }
        hs.shapes.push_back(shape);
    }

    void add_mesh(GdptShape &shape, HostMesh &&mesh) {
        shape.type = GDPT_SHAPE_TRIMESH;
        hs.meshes.push_back(std::move(mesh));
        // pointers are filled by HostScene::finalize()
        shape.num_vertices = (int)(hs.meshes.back().positions.size() / 3);
        shape.num_triangles = (int)(hs.meshes.back().indices.size() / 3);
        shape._pad = (int)hs.meshes.size() - 1; // mesh slot, resolved in finalize()
    }

    void parse_sensor(const XmlNode &node) {
        double fov = 45.0;
        M4 to_world = M4::identity();
        int width = 256, height = 256;
        int filter_type = GDPT_FILTER_BOX;
        double filter_param = 1.0;
        enum { AX, AY, ADIAG, ASMALL, ALARGE } axis = AX;
        std::string filename = "image.exr";
        if (node.attr("type") != "perspective") fail("Unsupported sensor: " + node.attr("type"));
        for (auto &cp : node.children) {
            const XmlNode &c = *cp;
            std::string name = c.attr("name");
            if (name == "fov") fov = parse_float(c.attr("value"), dm);
            else if (name == "toWorld" || name == "to_world") to_world = parse_transform(c, dm);
            else if (name == "fovAxis" || name == "fov_axis") {
                std::string v = c.attr("value");
                if (v == "x") axis = AX; else if (v == "y") axis = AY; else if (v == "diagonal") axis = ADIAG;
                else if (v == "smaller") axis = ASMALL; else if (v == "larger") axis = ALARGE;
                else fail("Unknown fovAxis value: " + v);
            }
        }
        for (auto &cp : node.children) {
            const XmlNode &c = *cp;
            if (c.name == "film") {
                // parse_film resets to its defaults on every <film> (parse_scene.cpp:604-609)
                width = height = 256; filename = "image.exr"; filter_type = GDPT_FILTER_BOX; filter_param = 1.0;
                for (auto &gp : c.children) {
                    const XmlNode &g = *gp;
                    std::string name = g.attr("name");
                    if (name == "width") width = parse_integer(g.attr("value"), dm);
                    else if (name == "height") height = parse_integer(g.attr("value"), dm);
                    else if (name == "filename") filename = parse_string(g.attr("value"), dm);
                    if (g.name == "rfilter") {
                        std::string ft = g.attr("type");
                        auto param = [&](const char *key, double def) {
                            double v = def;
                            for (auto &hp : g.children) if (hp->attr("name") == key) v = parse_float(hp->attr("value"), dm);
                            return v;
                        };
                        if (ft == "box") { filter_type = GDPT_FILTER_BOX; filter_param = param("width", 1.0); }
                        else if (ft == "tent") { filter_type = GDPT_FILTER_TENT; filter_param = param("width", 2.0); }
                        else if (ft == "gaussian") { filter_type = GDPT_FILTER_GAUSSIAN; filter_param = param("stddev", 0.5); }
                    }
                }
            } else if (c.name == "sampler") {
                if (c.attr("type") != "independent")
                    std::cerr << "Warning: the renderer currently only supports independent samplers." << std::endl;
                for (auto &gp : c.children) {
                    std::string name = gp->attr("name");
                    if (name == "sampleCount" || name == "sample_count")
                        hs.desc.samples_per_pixel = parse_integer(gp->attr("value"), dm);
                }
            }
        }
        if (film_w > 0) width = film_w;
        if (film_h > 0) height = film_h;
        // to fovX (parse_scene.cpp:842-855)
        if (axis == AY || (axis == ASMALL && height < width) || (axis == ALARGE && width < height)) {
            double aspect = width / (double)height;
            fov = degrees(2 * std::atan(std::tan(radians(fov) / 2) * aspect));
        } else if (axis == ADIAG) {
            double aspect = width / (double)height;
            double diagonal = 2 * std::tan(radians(fov) / 2);
            double w = diagonal / std::sqrt(1 + 1 / (aspect * aspect));
            fov = degrees(2 * std::atan(w / 2));
        }
        make_camera(to_world, fov, width, height, filter_type, filter_param, &hs.desc.camera);
        std::snprintf(hs.desc.output_filename, sizeof(hs.desc.output_filename), "%s", filename.c_str());
    }

    void parse_integrator(const XmlNode &node) {
        std::string type = node.attr("type");
        hs.desc.max_depth = -1; hs.desc.rr_depth = 5; // RenderOptions defaults, src/scene.h:25-32
        auto depth_args = [&]() {
            for (auto &cp : node.children) {
                std::string name = cp->attr("name");
                if (name == "maxDepth") hs.desc.max_depth = parse_integer(cp->attr("value"), dm);
                else if (name == "rrDepth") hs.desc.rr_depth = parse_integer(cp->attr("value"), dm);
            }
        };
        if (type == "path") { hs.desc.integrator = GDPT_INTEGRATOR_PATH; depth_args(); }
        else if (type == "gradpath") { hs.desc.integrator = GDPT_INTEGRATOR_GRADPATH; depth_args(); }
        else if (type == "direct") { hs.desc.integrator = GDPT_INTEGRATOR_PATH; hs.desc.max_depth = 2; }
        else if (type == "volpath" || type == "depth" || type == "shadingNormal" || type == "shading_normal" ||
                 type == "meanCurvature" || type == "mean_curvature" || type == "rayDifferential" ||
                 type == "ray_differential" || type == "mipmapLevel" || type == "mipmap_level")
            hs.desc.integrator = GDPT_INTEGRATOR_OTHER;
        else fail("Unsupported integrator: " + type);
    }

    void run(const XmlNode &scene) {
        // defaults of parse_scene(): RenderOptions{} and a 45-degree 256x256 box-filtered camera
        hs.desc.integrator = GDPT_INTEGRATOR_PATH;
        hs.desc.samples_per_pixel = 4; hs.desc.max_depth = -1; hs.desc.rr_depth = 5;
        make_camera(M4::identity(), 45.0, 256, 256, GDPT_FILTER_BOX, 1.0, &hs.desc.camera);
        std::snprintf(hs.desc.output_filename, sizeof(hs.desc.output_filename), "image.exr");
        for (auto &cp : scene.children) {
            const XmlNode &c = *cp;
            if (c.name == "default") {
                if (c.has_attr("name") && c.has_attr("value")) dm[c.attr("name")] = c.attr("value");
            } else if (c.name == "integrator") {
                parse_integrator(c);
            } else if (c.name == "sensor") {
                hs.desc.samples_per_pixel = 4; // ParsedSampler default, overwritten if <sampler> present
                parse_sensor(c);
            } else if (c.name == "bsdf") {
                auto [mname, m] = parse_bsdf(c);
                if (!mname.empty()) { material_map[mname] = (int)hs.materials.size(); hs.materials.push_back(m); }
            } else if (c.name == "shape") {
                parse_shape(c);
            } else if (c.name == "texture") {
                std::string id = c.attr("id");
                if (texture_map.count(id)) fail("Duplicated texture ID:" + id);
                texture_map[id] = parse_texture(c, dm);
            } else if (c.name == "emitter") {
                std::string type = c.attr("type");
                if (type == "envmap") {
                    // src/parsers/parse_scene.cpp:1484-1508. Integrator::GradPath never evaluates the environment
                    // (src/path_tracing.h:982-985); Integrator::Path samples it and looks it up.
                    std::string filename;
                    double scale = 1;
                    M4 to_world = M4::identity();
                    for (auto &gp : c.children) {
                        const XmlNode &g = *gp;
                        std::string n = g.attr("name");
                        if (n == "filename") filename = parse_string(g.attr("value"), dm);
                        else if (n == "toWorld" || n == "to_world") to_world = parse_transform(g, dm);
                        else if (n == "scale") scale = parse_float(g.attr("value"), dm);
                    }
                    if (filename.empty()) fail("Filename unspecified for envmap.");
                    GdptEnvmap &e = hs.desc.envmap;
                    e.image_id = insert_image("__envmap_texture__", filename, 3);
                    e.scale = scale;
                    M4 to_local = inverse(to_world);
                    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { e.to_world[4 * i + j] = to_world(i, j); e.to_local[4 * i + j] = to_local(i, j); }
                    e.light_id = (int)hs.lights.size();
                    GdptLight l{};
                    l.shape_id = -1;
                    hs.lights.push_back(l);
                    hs.desc.has_envmap = 1;
                } else if (type == "point" || type == "directional") {
                    fail("emitter type '" + type + "' is outside the GradPath hot-path subset");
                } else {
                    fail("Unknown emitter type:" + type);
                }
            }
            // <medium>: volumetric integrators only, ignored here
        }
    }
};

} // namespace

// ---------------------------------------------------------------------------

void HostScene::finalize() {
    for (auto &s : shapes) {
        if (s.type == GDPT_SHAPE_TRIMESH) {
            const HostMesh &m = meshes[(size_t)s._pad];
            s.positions = m.positions.data();
            s.indices = m.indices.data();
            s.normals = m.normals.empty() ? nullptr : m.normals.data();
            s.uvs = m.uvs.empty() ? nullptr : m.uvs.data();
        }
    }
    desc.num_materials = (int)materials.size(); desc.materials = materials.data();
    desc.num_shapes = (int)shapes.size(); desc.shapes = shapes.data();
    desc.num_lights = (int)lights.size(); desc.lights = lights.data();
    desc.num_images = (int)images.size(); desc.images = images.data();
}

void make_camera(const M4 &cam_to_world, double fov_deg, int width, int height,
                 int filter_type, double filter_param, GdptCamera *out) {
    double aspect = (double)width / (double)height;
    M4 cam_to_sample = scale(V3{-0.5, -0.5 * aspect, 1.0}) * translate(V3{-1.0, -1.0 / aspect, 0.0}) * perspective(fov_deg);
    M4 sample_to_cam = inverse(cam_to_sample);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            out->sample_to_cam[i * 4 + j] = sample_to_cam(i, j);
            out->cam_to_world[i * 4 + j] = cam_to_world(i, j);
        }
    out->width = width; out->height = height;
    out->filter_type = filter_type; out->filter_param = filter_param;
}

// CIE 1931 fits and the 1 nm Riemann sum of src/spectrum.h:48-111, then XYZ->RGB (:113-118)
V3 spectrum_to_rgb(const std::vector<std::pair<double, double>> &data) {
    auto xfit = [](double w) {
        double t1 = (w - 442.0) * ((w < 442.0) ? 0.0624 : 0.0374);
        double t2 = (w - 599.8) * ((w < 599.8) ? 0.0264 : 0.0323);
        double t3 = (w - 501.1) * ((w < 501.1) ? 0.0490 : 0.0382);
        return 0.362 * std::exp(-0.5 * t1 * t1) + 1.056 * std::exp(-0.5 * t2 * t2) - 0.065 * std::exp(-0.5 * t3 * t3);
    };
    auto yfit = [](double w) {
        double t1 = (w - 568.8) * ((w < 568.8) ? 0.0213 : 0.0247);
        double t2 = (w - 530.9) * ((w < 530.9) ? 0.0613 : 0.0322);
        return 0.821 * std::exp(-0.5 * t1 * t1) + 0.286 * std::exp(-0.5 * t2 * t2);
    };
    auto zfit = [](double w) {
        double t1 = (w - 437.0) * ((w < 437.0) ? 0.0845 : 0.0278);
        double t2 = (w - 459.0) * ((w < 459.0) ? 0.0385 : 0.0725);
        return 1.217 * std::exp(-0.5 * t1 * t1) + 0.681 * std::exp(-0.5 * t2 * t2);
    };
    const double cie_y_integral = 106.856895, wl_beg = 400, wl_end = 700;
    if (data.empty()) return {0, 0, 0};
    V3 ret{0, 0, 0};
    int pos = 0, n = (int)data.size();
    for (double wl = wl_beg; wl <= wl_end; wl += 1.0) {
        while (pos < n - 1 && !((data[pos].first <= wl && data[pos + 1].first > wl) || data[0].first > wl)) pos += 1;
        double meas;
        if (pos < n - 1 && data[0].first <= wl) {
            int nx = std::min(pos + 1, n - 1);
            double cd = data[pos].second, nd = data[nx].second, cw = data[pos].first, nw = data[nx].first;
            meas = cd * (nw - wl) / (nw - cw) + nd * (wl - cw) / (nw - cw);
        } else {
            meas = data[pos].second;
        }
        V3 coeff{xfit(wl), yfit(wl), zfit(wl)};
        ret = ret + coeff * meas;
    }
    double span = wl_end - wl_beg;
    ret = ret * (span / (cie_y_integral * (wl_end - wl_beg)));
    return xyz_to_rgb(ret);
}

std::vector<double> compute_vertex_normals(const std::vector<double> &positions, const std::vector<int32_t> &indices) {
    size_t nv = positions.size() / 3;
    std::vector<V3> normals(nv, V3{0, 0, 0});
    auto P = [&](int i) { return V3{positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]}; };
    auto unit_angle = [](const V3 &u, const V3 &v) {
        if (dot(u, v) < 0) return (kPi - 2) * std::asin(0.5 * length(v + u)); // sic: src/parsers/shape_utils.h:9-10
        return 2 * std::asin(0.5 * length(v - u));
    };
    for (size_t t = 0; t + 2 < indices.size(); t += 3) {
        const int32_t *idx = &indices[t];
        V3 n{0, 0, 0};
        for (int i = 0; i < 3; i++) {
            V3 v0 = P(idx[i]), v1 = P(idx[(i + 1) % 3]), v2 = P(idx[(i + 2) % 3]);
            V3 s1 = v1 - v0, s2 = v2 - v0;
            if (i == 0) {
                n = cross(s1, s2);
                double l = length(n);
                if (l == 0) break;
                n = n / l;
            }
            double angle = unit_angle(normalize(s1), normalize(s2));
            normals[idx[i]] = normals[idx[i]] + n * angle;
        }
    }
    std::vector<double> out(nv * 3);
    for (size_t i = 0; i < nv; i++) {
        V3 n = normals[i];
        double l = length(n);
        n = (l != 0) ? n / l : V3{0, 0, 0};
        out[3 * i] = n.x; out[3 * i + 1] = n.y; out[3 * i + 2] = n.z;
    }
    return out;
}

HostMesh load_obj(const std::string &filename, const M4 &to_world) {
    std::ifstream ifs(filename, std::ifstream::in);
    if (!ifs.is_open()) fail("Unable to open the obj file");
    std::vector<V3> pos_pool, nor_pool;
    std::vector<V2> st_pool;
    struct Key { int v, vt, vn; bool operator<(const Key &o) const { return v != o.v ? v < o.v : (vt != o.vt ? vt < o.vt : vn < o.vn); } };
    std::map<Key, int> vmap;
    HostMesh mesh;
    M4 inv_world = inverse(to_world);

    auto face_ids = [](const std::string &s) {
        std::vector<int> r;
        for (auto &f : split_slash(s)) r.push_back(f.empty() ? 0 : to_int(f));
        while (r.size() < 3) r.push_back(0);
        return Key{r[0] - 1, r[1] - 1, r[2] - 1};
    };
    auto vertex_id = [&](const Key &k) {
        auto it = vmap.find(k);
        if (it != vmap.end()) return it->second;
        int id = (int)(mesh.positions.size() / 3);
        if (k.v < 0 || k.v >= (int)pos_pool.size()) fail("obj: vertex index out of range in " + filename);
        V3 p = xform_point(to_world, pos_pool[k.v]);
        mesh.positions.insert(mesh.positions.end(), {p.x, p.y, p.z});
        if (k.vt != -1) {
            if (k.vt < 0 || k.vt >= (int)st_pool.size()) fail("obj: uv index out of range in " + filename);
            mesh.uvs.insert(mesh.uvs.end(), {st_pool[k.vt].x, st_pool[k.vt].y});
        }
        if (k.vn != -1) {
            if (k.vn < 0 || k.vn >= (int)nor_pool.size()) fail("obj: normal index out of range in " + filename);
            V3 n = xform_normal(inv_world, nor_pool[k.vn]);
            mesh.normals.insert(mesh.normals.end(), {n.x, n.y, n.z});
        }
        vmap[k] = id;
        return id;
    };

    std::string line;
    while (ifs.good()) {
        std::getline(ifs, line);
        size_t b = 0, e = line.size();
        while (b < e && std::isspace((unsigned char)line[b])) b++;
        while (e > b && std::isspace((unsigned char)line[e - 1])) e--;
        line = line.substr(b, e - b);
        if (line.empty() || line[0] == '#') continue;
        std::stringstream ss(line);
        std::string token;
        ss >> token;
        if (token == "v") {
            double x = 0, y = 0, z = 0, w = 1;
            ss >> x >> y >> z >> w;
            pos_pool.push_back(V3{x, y, z} / w);
        } else if (token == "vt") {
            double s = 0, t = 0, w = 0;
            ss >> s >> t >> w;
            st_pool.push_back(V2{s, 1 - t});
        } else if (token == "vn") {
            double x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            nor_pool.push_back(normalize(V3{x, y, z}));
        } else if (token == "f") {
            std::string i0, i1, i2, i3, i4;
            ss >> i0 >> i1 >> i2;
            int a = vertex_id(face_ids(i0)), bq = vertex_id(face_ids(i1)), c = vertex_id(face_ids(i2));
            mesh.indices.insert(mesh.indices.end(), {a, bq, c});
            if (ss >> i3) {
                int d = vertex_id(face_ids(i3));
                mesh.indices.insert(mesh.indices.end(), {a, c, d}); // quad -> (0,1,2),(0,2,3)
            }
            if (ss >> i4) fail("The object file contains n-gon (n>4) that we do not support.");
        }
    }
    // the reference keeps uvs/normals only when every vertex pushed one; a ragged array would be
    // indexed out of range there (undefined) — refuse instead
    size_t nv = mesh.positions.size() / 3;
    if (!mesh.uvs.empty() && mesh.uvs.size() != nv * 2) fail("obj: only some vertices carry uvs: " + filename);
    if (!mesh.normals.empty() && mesh.normals.size() != nv * 3) fail("obj: only some vertices carry normals: " + filename);
    return mesh;
}

HostMesh load_serialized(const std::string &filename, int shape_index, const M4 &to_world) {
    std::ifstream f(filename, std::ios::binary);
    if (!f.is_open()) fail("Unable to open the serialized file " + filename);
    std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (file.size() < 8) fail("serialized: file too short: " + filename);
    auto rd16 = [&](size_t off) { uint16_t v; std::memcpy(&v, &file[off], 2); return v; };
    uint16_t version = rd16(2);
    size_t offset = 4; // after format id + version
    if (shape_index > 0) {
        uint32_t count;
        std::memcpy(&count, &file[file.size() - 4], 4);
        if ((uint32_t)shape_index >= count) fail("serialized: shape index out of range");
        const size_t entry = (version == 4) ? 8 : 4;
        if ((size_t)count > (file.size() - 4) / entry) fail("serialized: corrupt shape table in " + filename);
        size_t sub = 0;
        if (version == 4) {
            uint64_t o;
            std::memcpy(&o, &file[file.size() - 4 - 8 * (size_t)(count - shape_index)], 8);
            sub = (size_t)o;
        } else {
            uint32_t o;
            std::memcpy(&o, &file[file.size() - 4 * (size_t)(count - shape_index + 1)], 4);
            sub = o;
        }
        if (sub >= file.size() - 4) fail("serialized: bad offset");
        offset = sub + 4;
    }
    if (offset >= file.size()) fail("serialized: bad offset");
    // inflate the zlib stream that starts at `offset`
    z_stream zs{};
    if (inflateInit2(&zs, 15) != Z_OK) fail("Could not initialize ZLIB");
    zs.next_in = &file[offset];
    zs.avail_in = (uInt)std::min<size_t>(file.size() - offset, 0xFFFFFFFFu);
    std::vector<unsigned char> out;
    unsigned char buf[1 << 16];
    int ret = Z_OK;
    while (ret != Z_STREAM_END) {
        zs.next_out = buf; zs.avail_out = sizeof(buf);
        ret = inflate(&zs, Z_NO_FLUSH);
        if (ret != Z_OK && ret != Z_STREAM_END) { inflateEnd(&zs); fail("inflate(): data error!"); }
        out.insert(out.end(), buf, buf + (sizeof(buf) - zs.avail_out));
        if (ret == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) { inflateEnd(&zs); fail("Read less data than expected"); }
    }
    inflateEnd(&zs);

    size_t p = 0;
    auto need = [&](size_t n) { if (p + n > out.size()) fail("inflate(): attempting to read past the end of the stream!"); };
    auto rd = [&](void *dst, size_t n) { need(n); std::memcpy(dst, &out[p], n); p += n; };
    uint32_t flags; rd(&flags, 4);
    if (version == 4) { char c; do { rd(&c, 1); } while (c != '\0'); }
    uint64_t nv, nt; rd(&nv, 8); rd(&nt, 8);
    bool dbl = flags & 0x2000;
    // every vertex carries at least a position, every triangle three 32-bit indices: counts beyond what the inflated
    // stream can hold are corrupt (and must not reach the allocations below)
    if (nv > out.size() / (dbl ? 24 : 12) || nt > out.size() / 12) fail("serialized: corrupt vertex / triangle count in " + filename);
    auto rdreal = [&]() { if (dbl) { double v; rd(&v, 8); return v; } float v; rd(&v, 4); return (double)v; };
    HostMesh mesh;
    mesh.positions.resize(nv * 3);
    for (uint64_t i = 0; i < nv; i++) {
        V3 q; q.x = rdreal(); q.y = rdreal(); q.z = rdreal();
        q = xform_point(to_world, q);
        mesh.positions[3 * i] = q.x; mesh.positions[3 * i + 1] = q.y; mesh.positions[3 * i + 2] = q.z;
    }
    if (flags & 0x0001) {
        M4 inv = inverse(to_world);
        mesh.normals.resize(nv * 3);
        for (uint64_t i = 0; i < nv; i++) {
            V3 n; n.x = rdreal(); n.y = rdreal(); n.z = rdreal();
            n = xform_normal(inv, n);
            mesh.normals[3 * i] = n.x; mesh.normals[3 * i + 1] = n.y; mesh.normals[3 * i + 2] = n.z;
        }
    }
    if (flags & 0x0002) {
        mesh.uvs.resize(nv * 2);
        for (uint64_t i = 0; i < nv * 2; i++) mesh.uvs[i] = rdreal();
    }
    if (flags & 0x0008) for (uint64_t i = 0; i < nv * 3; i++) (void)rdreal();
    mesh.indices.resize(nt * 3);
    for (uint64_t i = 0; i < nt * 3; i++) { int32_t v; rd(&v, 4); mesh.indices[i] = v; }
    return mesh;
}

std::unique_ptr<HostScene> load_scene_xml(const std::string &path, int film_width, int film_height) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) fail("Parse error: cannot open " + path);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    XmlParser parser(text);
    std::unique_ptr<XmlNode> doc = parser.parse_document();
    const XmlNode *scene = doc->child("scene");
    auto hs = std::make_unique<HostScene>();
    Builder b(*hs);
    // the reference chdir()s into the scene's folder while parsing (parse_scene.cpp:1624-1628);
    // resolve relative asset paths against it instead of changing the process cwd
    b.scene_dir = fs::path(path).parent_path();
    b.film_w = film_width; b.film_h = film_height;
    if (scene) b.run(*scene);
    else b.run(XmlNode{}); // pugixml yields an empty node: defaults everywhere
    hs->finalize();
    // index validation (the reference would index out of range later)
    for (auto &s : hs->shapes) {
        if (s.material_id < 0 || s.material_id >= (int)hs->materials.size())
            fail("a shape has no material (the reference indexes scene.materials[-1] here)");
        if (s.type == GDPT_SHAPE_TRIMESH)
            for (int i = 0; i < s.num_triangles * 3; i++)
                if (s.indices[i] < 0 || s.indices[i] >= s.num_vertices) fail("mesh index out of range");
    }
    return hs;
}

} // namespace gdpt
