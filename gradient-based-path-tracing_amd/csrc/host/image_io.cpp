// image_io.cpp — see image_io.h. Own writers (the reference vendors tinyexr/stb_image).
#include "image_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace gdpt {

namespace {

bool ends_with(const std::string &s, const std::string &suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// IEEE binary32 -> binary16, round to nearest even, overflow to inf, NaN kept.
uint16_t float_to_half(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7FFFFFFFu;
    if (em >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (em > 0x7F800000u ? 0x200u | ((em >> 13) & 0x3FFu) : 0u));
    if (em >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);             // >= 65536 -> inf (values rounding up to it handled below)
    if (em < 0x33000001u) return (uint16_t)sign;                           // < 2^-25 rounds to zero
    int exp = (int)(em >> 23) - 127;
    uint32_t man = (em & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (exp < -14) { shift = 13 + (-14 - exp); hexp = 0; } else { shift = 13; hexp = (uint32_t)(exp + 15); }
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
    uint32_t h;
    if (hexp == 0) h = half_man;                       // subnormal (may carry into exponent 1 naturally)
    else h = ((hexp << 10) + (half_man - 0x400u));     // carry propagates into the exponent
    if (h >= 0x7C00u) h = 0x7C00u;
    return (uint16_t)(sign | h);
}

void put32(std::vector<unsigned char> &b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((unsigned char)(v >> (8 * i))); }
void put64(std::vector<unsigned char> &b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((unsigned char)(v >> (8 * i))); }
void putstr(std::vector<unsigned char> &b, const char *s) { while (*s) b.push_back((unsigned char)*s++); b.push_back(0); }
void attr(std::vector<unsigned char> &b, const char *name, const char *type, const std::vector<unsigned char> &val) {
    putstr(b, name); putstr(b, type); put32(b, (uint32_t)val.size());
    b.insert(b.end(), val.begin(), val.end());
}

void write_exr_half(const std::string &filename, int w, int h, const double *rgb) {
    std::vector<unsigned char> b;
    put32(b, 20000630u);  // magic
    put32(b, 2u);         // version 2, single-part scanline
    {   // channels: B, G, R (alphabetical), HALF, linear, sampling 1,1
        std::vector<unsigned char> v;
        for (const char *c : {"B", "G", "R"}) {
            putstr(v, c); put32(v, 1u /*HALF*/); v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);
            put32(v, 1); put32(v, 1);
        }
        v.push_back(0);
        attr(b, "channels", "chlist", v);
    }
    { std::vector<unsigned char> v{0 /*NO_COMPRESSION*/}; attr(b, "compression", "compression", v); }
    { std::vector<unsigned char> v; put32(v, 0); put32(v, 0); put32(v, (uint32_t)(w - 1)); put32(v, (uint32_t)(h - 1));
      attr(b, "dataWindow", "box2i", v); attr(b, "displayWindow", "box2i", v); }
    { std::vector<unsigned char> v{0 /*INCREASING_Y*/}; attr(b, "lineOrder", "lineOrder", v); }
    { std::vector<unsigned char> v; float one = 1.0f; uint32_t u; std::memcpy(&u, &one, 4); put32(v, u); attr(b, "pixelAspectRatio", "float", v); }
    { std::vector<unsigned char> v; put32(v, 0); put32(v, 0); attr(b, "screenWindowCenter", "v2f", v); }
    { std::vector<unsigned char> v; float one = 1.0f; uint32_t u; std::memcpy(&u, &one, 4); put32(v, u); attr(b, "screenWindowWidth", "float", v); }
    b.push_back(0); // end of header
    size_t table = b.size();
    size_t line_bytes = (size_t)w * 3 * 2;
    size_t first = table + (size_t)h * 8;
    for (int y = 0; y < h; y++) put64(b, (uint64_t)(first + (size_t)y * (8 + line_bytes)));
    for (int y = 0; y < h; y++) {
        put32(b, (uint32_t)y);
        put32(b, (uint32_t)line_bytes);
        for (int c : {2, 1, 0}) // B, G, R planes
            for (int x = 0; x < w; x++) {
                uint16_t hv = float_to_half((float)rgb[((size_t)y * w + x) * 3 + c]);
                b.push_back((unsigned char)(hv & 0xFF)); b.push_back((unsigned char)(hv >> 8));
            }
    }
    std::ofstream ofs(filename, std::ios::binary);
    if (!ofs) throw std::runtime_error("Failure when writing image: " + filename);
    ofs.write((const char *)b.data(), (std::streamsize)b.size());
}

} // namespace

void write_image(const std::string &filename, int width, int height, const double *rgb) {
    if (ends_with(filename, ".pfm")) {
        std::ofstream ofs(filename, std::ios::binary);
        if (!ofs) throw std::runtime_error("Failure when writing image: " + filename);
        ofs << "PF" << std::endl << width << " " << height << std::endl << "-1" << std::endl;
        std::vector<float> data((size_t)width * height * 3);
        for (size_t i = 0; i < data.size(); i++) data[i] = (float)rgb[i];
        ofs.write((const char *)data.data(), (std::streamsize)(data.size() * sizeof(float)));
    } else if (ends_with(filename, ".exr")) {
        write_exr_half(filename, width, height, rgb);
    }
}

void read_pfm(const std::string &filename, int *width, int *height, std::vector<double> *rgb) {
    std::ifstream ifs(filename, std::ios::binary);
    if (!ifs) throw std::runtime_error("Failure when loading image: " + filename);
    std::string magic;
    double scale;
    ifs >> magic >> *width >> *height >> scale;
    ifs.get(); // single whitespace after the scale
    int ch = (magic == "PF") ? 3 : (magic == "Pf" ? 1 : 0);
    if (!ch || *width <= 0 || *height <= 0) throw std::runtime_error("Unsupported image format: " + filename);
    std::vector<float> data((size_t)*width * *height * ch);
    ifs.read((char *)data.data(), (std::streamsize)(data.size() * sizeof(float)));
    if (!ifs) throw std::runtime_error("Failure when loading image: " + filename);
    rgb->resize((size_t)*width * *height * 3);
    for (size_t i = 0; i < (size_t)*width * *height; i++)
        for (int c = 0; c < 3; c++) (*rgb)[i * 3 + c] = data[i * ch + (ch == 3 ? c : 0)];
}

void load_texture_file(const std::string &path, int channels, int *width, int *height, std::vector<double> *texels) {
    if (ends_with(path, ".pfm")) {
        std::vector<double> rgb;
        read_pfm(path, width, height, &rgb);
        size_t n = (size_t)*width * *height;
        if (channels == 3) { *texels = std::move(rgb); return; }
        texels->resize(n);
        for (size_t i = 0; i < n; i++) (*texels)[i] = (rgb[3 * i] + rgb[3 * i + 1] + rgb[3 * i + 2]) / 3;
        return;
    }
    std::string raw = path + ".gdtex";
    std::ifstream ifs(raw, std::ios::binary);
    if (!ifs) throw std::runtime_error("Failure when loading image: " + path +
                                       " (no image codec on the hot path: expected the pre-decoded companion " + raw + ")");
    char magic[7];
    int32_t hdr[3];
    ifs.read(magic, 7);
    ifs.read((char *)hdr, 12);
    bool v1 = std::memcmp(magic, "GDTEX1\n", 7) == 0, v2 = std::memcmp(magic, "GDTEX2\n", 7) == 0;
    if (!ifs || !(v1 || v2) || hdr[0] <= 0 || hdr[1] <= 0 || (hdr[2] != 1 && hdr[2] != 3))
        throw std::runtime_error("Failure when loading image: bad header in " + raw);
    *width = hdr[0]; *height = hdr[1];
    int fc = hdr[2];
    std::vector<float> data((size_t)hdr[0] * hdr[1] * fc);
    if (v1) {
        ifs.read((char *)data.data(), (std::streamsize)(data.size() * sizeof(float)));
        if (!ifs) throw std::runtime_error("Failure when loading image: truncated " + raw);
    } else {
        // 8-bit texels, zlib-compressed; widened like stbi_loadf: (float)pow(v/255.0f, 2.2f) (stb_image.h:1849).
        // A 1-channel request is reduced in 8 bits first, as stb does: y = (77 r + 150 g + 29 b) >> 8 (stb_image.h:1708).
        uint32_t zlen = 0;
        ifs.read((char *)&zlen, 4);
        std::vector<unsigned char> z(zlen), u8((size_t)hdr[0] * hdr[1] * fc);
        ifs.read((char *)z.data(), zlen);
        if (!ifs) throw std::runtime_error("Failure when loading image: truncated " + raw);
        uLongf out_len = (uLongf)u8.size();
        if (uncompress(u8.data(), &out_len, z.data(), zlen) != Z_OK || out_len != u8.size())
            throw std::runtime_error("Failure when loading image: corrupt " + raw);
        float lut[256];
        for (int v = 0; v < 256; v++) lut[v] = (float)(std::pow((double)(v / 255.0f), (double)2.2f) * (double)1.0f);
        size_t npx = (size_t)hdr[0] * hdr[1];
        if (channels == 1 && fc == 3) {
            data.resize(npx);
            for (size_t i = 0; i < npx; i++) {
                int y = (u8[3 * i] * 77 + u8[3 * i + 1] * 150 + 29 * u8[3 * i + 2]) >> 8;
                data[i] = lut[y & 255];
            }
            fc = 1;
        } else {
            for (size_t i = 0; i < u8.size(); i++) data[i] = lut[u8[i]];
        }
    }
    size_t n = (size_t)hdr[0] * hdr[1];
    texels->resize(n * channels);
    for (size_t i = 0; i < n; i++) {
        if (channels == 3) for (int c = 0; c < 3; c++) (*texels)[3 * i + c] = data[i * fc + (fc == 3 ? c : 0)];
        else (*texels)[i] = (fc == 1) ? data[i] : (data[3 * i] + data[3 * i + 1] + data[3 * i + 2]) / 3;
    }
}

} // namespace gdpt
