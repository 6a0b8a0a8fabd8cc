// image_io.cpp — see image_io.h. Own writers (the reference vendors tinyexr/stb_image).
#include "image_io.h"
#include "jpeg_decode.h"
#include "png_decode.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <iterator>
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

// binary32 -> binary16 the way the reference's writer does it (tinyexr's conversion behind SaveEXR,
// src/3rdparty/tinyexr.h:889-924): the first dropped mantissa bit alone decides the rounding (ties go away from
// zero, not to even), a carry may run into the exponent (up to infinity), float subnormals flush to signed zero,
// NaN becomes the quiet NaN 0x7E00 pattern.
uint16_t float_to_half(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, e = (x >> 23) & 0xFFu, m = x & 0x7FFFFFu;
    uint32_t h = 0;
    if (e == 0) h = 0;
    else if (e == 255) h = 0x7C00u | (m ? 0x200u : 0u);
    else {
        const int ne = (int)e - 127 + 15;
        if (ne >= 31) h = 0x7C00u;
        else if (ne <= 0) {
            if (14 - ne <= 24) {
                const uint32_t mant = m | 0x800000u;
                h = mant >> (14 - ne);
                if ((mant >> (13 - ne)) & 1u) h++;
            }
        } else {
            h = ((uint32_t)ne << 10) | (m >> 13);
            if (m & 0x1000u) h++;
        }
    }
    return (uint16_t)(sign | h);
}

void put32(std::vector<unsigned char> &b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((unsigned char)(v >> (8 * i))); }
void put64(std::vector<unsigned char> &b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((unsigned char)(v >> (8 * i))); }
void putstr(std::vector<unsigned char> &b, const char *s) { while (*s) b.push_back((unsigned char)*s++); b.push_back(0); }
void attr(std::vector<unsigned char> &b, const char *name, const char *type, const std::vector<unsigned char> &val) {
    putstr(b, name); putstr(b, type); put32(b, (uint32_t)val.size());
    b.insert(b.end(), val.begin(), val.end());
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {                                   // subnormal: normalise
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7F800000u | (man << 13);
    else x = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

// OpenEXR ZIP block coding: bytes split into even/odd halves, delta-predicted, deflated; stored raw if that is not
// smaller (OpenEXR file layout document, "ZIP_COMPRESSION"; the reference gets it from tinyexr's SaveEXR).
std::vector<unsigned char> exr_zip_encode(const std::vector<unsigned char> &raw) {
    const size_t n = raw.size();
    std::vector<unsigned char> tmp(n);
    {
        unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (n + 1) / 2;
        for (size_t i = 0; i < n; i++) { if (i & 1) *t2++ = raw[i]; else *t1++ = raw[i]; }
    }
    {
        int p = tmp.empty() ? 0 : tmp[0];
        for (size_t i = 1; i < n; i++) { int d = (int)tmp[i] - p + (128 + 256); p = tmp[i]; tmp[i] = (unsigned char)d; }
    }
    uLongf zlen = compressBound((uLong)n);
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, tmp.data(), (uLong)n, Z_DEFAULT_COMPRESSION) != Z_OK) throw std::runtime_error("EXR: deflate failed");
    if (zlen >= n) return raw;
    z.resize(zlen);
    return z;
}
void exr_zip_decode(const unsigned char *src, size_t src_len, std::vector<unsigned char> &raw) {
    const size_t n = raw.size();
    if (src_len == n) { std::memcpy(raw.data(), src, n); return; }        // stored uncompressed
    std::vector<unsigned char> tmp(n);
    uLongf out_len = (uLongf)n;
    if (uncompress(tmp.data(), &out_len, src, (uLong)src_len) != Z_OK || out_len != n) throw std::runtime_error("EXR: inflate failed");
    for (size_t i = 1; i < n; i++) tmp[i] = (unsigned char)((int)tmp[i - 1] + (int)tmp[i] - 128);
    const unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (n + 1) / 2;
    for (size_t i = 0; i < n; i++) raw[i] = (i & 1) ? *t2++ : *t1++;
}
void exr_rle_decode(const unsigned char *src, size_t src_len, std::vector<unsigned char> &raw) {
    const size_t n = raw.size();
    if (src_len == n) { std::memcpy(raw.data(), src, n); return; }
    std::vector<unsigned char> tmp(n);
    size_t o = 0, i = 0;
    while (i < src_len) {
        int c = (signed char)src[i++];
        if (c < 0) { size_t cnt = (size_t)(-c); if (i + cnt > src_len || o + cnt > n) throw std::runtime_error("EXR: bad RLE"); std::memcpy(&tmp[o], &src[i], cnt); i += cnt; o += cnt; }
        else { size_t cnt = (size_t)c + 1; if (i >= src_len || o + cnt > n) throw std::runtime_error("EXR: bad RLE"); std::memset(&tmp[o], src[i++], cnt); o += cnt; }
    }
    if (o != n) throw std::runtime_error("EXR: bad RLE length");
    for (size_t k = 1; k < n; k++) tmp[k] = (unsigned char)((int)tmp[k - 1] + (int)tmp[k] - 128);
    const unsigned char *t1 = tmp.data(), *t2 = tmp.data() + (n + 1) / 2;
    for (size_t k = 0; k < n; k++) raw[k] = (k & 1) ? *t2++ : *t1++;
}

// ---- OpenEXR PIZ blocks (compression 4; the reference reads them through tinyexr, e.g. scenes/matpreview/envmap.exr) --
// Format (OpenEXR ImfPizCompressor / ImfHuf / ImfWav, as published): u16 minNonZero, u16 maxNonZero, the bytes
// [min..max] of a 8192-byte bitmap of the 16-bit values present, i32 length, then a Huffman stream — header im, iM,
// table length, nBits, reserved (5 x u32), the 6-bit code lengths of symbols im..iM with zero runs (63: 8-bit count + 6;
// 59..62: 2..5 zeros), canonical codes numbered from the longest length up, symbol iM = "repeat the previous value
// n times" (8-bit n) — of the wavelet-transformed, LUT-compacted 16-bit words laid out channel by channel; each
// channel (each 16-bit half of a 32-bit channel) is a 2-D hierarchical Haar-like transform, 14-bit or modulo-16-bit
// basis depending on the largest LUT index.
struct PizHuff {
    static constexpr int kSymbols = (1 << 16) + 1;
    int count[59] = {0};
    long long first[59] = {0};
    std::vector<int> order;        // symbols sorted by (length, index)
    int start[60] = {0};           // offset of each length's symbols in `order`
    void build(const std::vector<unsigned char> &len) {
        for (int l = 0; l < 59; l++) count[l] = 0;
        for (int s = 0; s < kSymbols; s++) count[len[(size_t)s]]++;
        long long c = 0;
        for (int l = 58; l > 0; l--) { first[l] = c; c = (c + count[l]) >> 1; }
        start[0] = 0;
        int acc = 0;
        for (int l = 1; l < 60; l++) { start[l] = acc; if (l < 59) acc += count[l]; }
        order.assign((size_t)acc, 0);
        std::vector<int> fill(start, start + 59);
        for (int s = 0; s < kSymbols; s++) { int l = len[(size_t)s]; if (l > 0) order[(size_t)fill[l]++] = s; }
    }
};
struct PizBits {
    const unsigned char *p, *end;
    unsigned long long acc = 0; int n = 0; long long budget;     // budget: bits of the stream still unread
    bool bit(int *b) {
        if (budget <= 0) return false;
        if (n == 0) { if (p >= end) return false; acc = *p++; n = 8; }
        *b = (int)((acc >> (n - 1)) & 1u); n--; budget--;
        return true;
    }
    bool bits(int k, unsigned *v) { unsigned r = 0; for (int i = 0; i < k; i++) { int b; if (!bit(&b)) return false; r = (r << 1) | (unsigned)b; } *v = r; return true; }
};
void piz_huf_uncompress(const unsigned char *src, size_t len, std::vector<uint16_t> &out) {
    auto fail = []() { throw std::runtime_error("EXR: corrupt PIZ Huffman data"); };
    if (len < 20) fail();
    auto u32 = [&](size_t o) { return (uint32_t)src[o] | ((uint32_t)src[o + 1] << 8) | ((uint32_t)src[o + 2] << 16) | ((uint32_t)src[o + 3] << 24); };
    const uint32_t im = u32(0), iM = u32(4), nbits = u32(12);
    if (im >= (uint32_t)PizHuff::kSymbols || iM >= (uint32_t)PizHuff::kSymbols || im > iM) fail();
    std::vector<unsigned char> lens((size_t)PizHuff::kSymbols, 0);
    PizBits tb{src + 20, src + len};
    tb.budget = (long long)(len - 20) * 8;
    for (uint32_t s = im; s <= iM; s++) {
        unsigned l;
        if (!tb.bits(6, &l)) fail();
        if (l == 63) { unsigned z; if (!tb.bits(8, &z)) fail(); z += 6; if (s + z > iM + 1) fail(); s += z - 1; }     // long zero run
        else if (l >= 59) { unsigned z = l - 59 + 2; if (s + z > iM + 1) fail(); s += z - 1; }                          // short zero run
        else lens[s] = (unsigned char)l;
    }
    PizHuff h;
    h.build(lens);
    const unsigned char *data = tb.p;           // the table is padded to a byte boundary
    if ((long long)nbits > (long long)(src + len - data) * 8) fail();
    PizBits br{data, src + len};
    br.budget = (long long)nbits;
    size_t o = 0;
    while (br.budget > 0 && o < out.size()) {
        long long code = 0;
        int sym = -1;
        for (int l = 1; l <= 58; l++) {
            int b;
            if (!br.bit(&b)) fail();
            code = (code << 1) | b;
            if (h.count[l] > 0 && code >= h.first[l] && code < h.first[l] + h.count[l]) { sym = h.order[(size_t)(h.start[l] + (int)(code - h.first[l]))]; break; }
        }
        if (sym < 0) fail();
        if ((uint32_t)sym == iM) {                          // run-length code
            unsigned run;
            if (!br.bits(8, &run) || o == 0 || o + run > out.size()) fail();
            const uint16_t prev = out[o - 1];
            for (unsigned i = 0; i < run; i++) out[o++] = prev;
        } else out[o++] = (uint16_t)sym;
    }
    if (o != out.size()) fail();
}
inline void piz_wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int ls = (int16_t)l, hi = (int16_t)h;
    const int ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hi);
}
inline void piz_wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xFFFF;
    const int aa = (d + bb - 0x8000) & 0xFFFF;
    b = (uint16_t)bb; a = (uint16_t)aa;
}
// Inverse of the PIZ 2-D wavelet. Source of the arithmetic: the OpenEXR file-format description of PIZ ("wav2Decode":
// hierarchical Haar-like lifting on 16-bit words, the 14-bit form when every value is < 2^14 and the modulo-2^16 form
// otherwise; piz_wdec14 / piz_wdec16 above are those two lifting steps and admit no other arithmetic). A level with
// half-step p pairs the samples that sit on the stride-p lattice: (x, x+p) for x = 0, 2p, ... while x + 2p <= nx, and
// the same in y; a lattice column (row) left without a partner — bit p of nx (ny) set — only takes part in the other
// direction. Cells of one level are independent, so the level is written here as two separable sweeps over the lattice
// (all vertical pairs, then all horizontal pairs) addressed by (x, y) indices, coarse to fine.
void piz_wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool narrow = mx < (1 << 14);
    auto unlift = [&](uint16_t &lo, uint16_t &hi) {          // (average, difference) -> the two samples, in place
        uint16_t a, b;
        if (narrow) piz_wdec14(lo, hi, a, b); else piz_wdec16(lo, hi, a, b);
        lo = a; hi = b;
    };
    auto at = [&](int x, int y) -> uint16_t & { return in[(ptrdiff_t)x * ox + (ptrdiff_t)y * oy]; };
    int top = 1;                                             // coarsest full step: largest power of two <= min(nx, ny)
    while (top * 2 <= (nx < ny ? nx : ny)) top *= 2;
    for (int step = top, half = top / 2; half >= 1; step = half, half /= 2) {
        const int cells_x = nx / step, cells_y = ny / step;  // complete cells per direction
        const bool odd_x = (nx & half) != 0, odd_y = (ny & half) != 0;
        // vertical pairs: both columns of every complete cell, plus the partnerless lattice column
        for (int cy = 0; cy < cells_y; cy++) {
            const int y = cy * step;
            for (int cx = 0; cx < cells_x; cx++) {
                unlift(at(cx * step, y), at(cx * step, y + half));
                unlift(at(cx * step + half, y), at(cx * step + half, y + half));
            }
            if (odd_x) unlift(at(cells_x * step, y), at(cells_x * step, y + half));
        }
        // horizontal pairs: both rows of every complete cell, plus the partnerless lattice row
        for (int cy = 0; cy < cells_y; cy++)
            for (int r = 0; r < 2; r++) {
                const int y = cy * step + r * half;
                for (int cx = 0; cx < cells_x; cx++) unlift(at(cx * step, y), at(cx * step + half, y));
            }
        if (odd_y) {
            const int y = cells_y * step;
            for (int cx = 0; cx < cells_x; cx++) unlift(at(cx * step, y), at(cx * step + half, y));
        }
    }
}
// chan_words[c]: 1 for HALF, 2 for FLOAT / UINT channels (in file order)
void exr_piz_decode(const unsigned char *src, size_t src_len, std::vector<unsigned char> &raw, const std::vector<int> &chan_words, int w, int lines) {
    if (src_len == raw.size()) { std::memcpy(raw.data(), src, src_len); return; }
    auto fail = []() { throw std::runtime_error("EXR: corrupt PIZ block"); };
    if (src_len < 4) fail();
    const unsigned min_nz = src[0] | (src[1] << 8), max_nz = src[2] | (src[3] << 8);
    size_t pos = 4;
    std::vector<unsigned char> bitmap(8192, 0);
    if (max_nz >= 8192) fail();
    if (min_nz <= max_nz) {
        if (pos + (max_nz - min_nz + 1) > src_len) fail();
        std::memcpy(&bitmap[min_nz], src + pos, max_nz - min_nz + 1);
        pos += max_nz - min_nz + 1;
    }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; i++) if (i == 0 || (bitmap[(size_t)i >> 3] & (1 << (i & 7)))) lut[(size_t)k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (pos + 4 > src_len) fail();
    const uint32_t hlen = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
    pos += 4;
    if (hlen > src_len - pos) fail();
    std::vector<uint16_t> tmp(raw.size() / 2);
    piz_huf_uncompress(src + pos, hlen, tmp);
    std::vector<size_t> chan_start(chan_words.size());
    size_t off = 0;
    for (size_t c = 0; c < chan_words.size(); c++) {
        chan_start[c] = off;
        for (int j = 0; j < chan_words[c]; j++) piz_wav2_decode(tmp.data() + off + j, w, chan_words[c], lines, w * chan_words[c], max_value);
        off += (size_t)w * lines * chan_words[c];
    }
    if (off != tmp.size()) fail();
    for (auto &v : tmp) v = lut[v];
    unsigned char *o = raw.data();
    std::vector<size_t> cursor = chan_start;
    for (int y = 0; y < lines; y++)
        for (size_t c = 0; c < chan_words.size(); c++) {
            const size_t n = (size_t)w * chan_words[c];
            std::memcpy(o, tmp.data() + cursor[c], n * 2);
            o += n * 2; cursor[c] += n;
        }
}

// RGB half-float scanline file, channels B, G, R. Like tinyexr's SaveEXR (which the reference calls with fp16 output,
// src/image.cpp:155-171): no compression when both extents are < 16, otherwise ZIP in 16-scanline blocks.
void write_exr_half(const std::string &filename, int w, int h, const double *rgb) {
    const bool zip = !(w < 16 && h < 16);
    const int lines_per_block = zip ? 16 : 1;
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
    { std::vector<unsigned char> v{(unsigned char)(zip ? 3 /*ZIP*/ : 0 /*NONE*/)}; attr(b, "compression", "compression", v); }
    { std::vector<unsigned char> v; put32(v, 0); put32(v, 0); put32(v, (uint32_t)(w - 1)); put32(v, (uint32_t)(h - 1));
      attr(b, "dataWindow", "box2i", v); attr(b, "displayWindow", "box2i", v); }
    { std::vector<unsigned char> v{0 /*INCREASING_Y*/}; attr(b, "lineOrder", "lineOrder", v); }
    { std::vector<unsigned char> v; float one = 1.0f; uint32_t u; std::memcpy(&u, &one, 4); put32(v, u); attr(b, "pixelAspectRatio", "float", v); }
    { std::vector<unsigned char> v; put32(v, 0); put32(v, 0); attr(b, "screenWindowCenter", "v2f", v); }
    { std::vector<unsigned char> v; float one = 1.0f; uint32_t u; std::memcpy(&u, &one, 4); put32(v, u); attr(b, "screenWindowWidth", "float", v); }
    b.push_back(0); // end of header
    const int nblocks = (h + lines_per_block - 1) / lines_per_block;
    const size_t table = b.size();
    b.resize(table + (size_t)nblocks * 8);
    for (int blk = 0; blk < nblocks; blk++) {
        const int y0 = blk * lines_per_block, y1 = std::min(h, y0 + lines_per_block);
        std::vector<unsigned char> raw;
        raw.reserve((size_t)(y1 - y0) * w * 6);
        for (int y = y0; y < y1; y++)
            for (int c : {2, 1, 0}) // B, G, R planes of the scanline
                for (int x = 0; x < w; x++) {
                    uint16_t hv = float_to_half((float)rgb[((size_t)y * w + x) * 3 + c]);
                    raw.push_back((unsigned char)(hv & 0xFF)); raw.push_back((unsigned char)(hv >> 8));
                }
        const std::vector<unsigned char> payload = zip ? exr_zip_encode(raw) : raw;
        const uint64_t off = b.size();
        for (int i = 0; i < 8; i++) b[table + (size_t)blk * 8 + i] = (unsigned char)(off >> (8 * i));
        put32(b, (uint32_t)y0);
        put32(b, (uint32_t)payload.size());
        b.insert(b.end(), payload.begin(), payload.end());
    }
    std::ofstream ofs(filename, std::ios::binary);
    if (!ofs) throw std::runtime_error("Failure when writing image: " + filename);
    ofs.write((const char *)b.data(), (std::streamsize)b.size());
}

// Scanline OpenEXR reader: single part, compression NONE / RLE / ZIPS / ZIP / PIZ, HALF / FLOAT / UINT channels.
// Returns what tinyexr's LoadEXR hands the reference (src/image.cpp:56-72,109-127): the R, G, B channels as fp32
// (a single-channel file is replicated to all three).
void read_exr_rgb(const std::string &filename, int *width, int *height, std::vector<float> *rgb) {
    std::ifstream ifs(filename, std::ios::binary);
    if (!ifs) throw std::runtime_error("Failure when loading image: " + filename);
    std::vector<unsigned char> f((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto need = [&](size_t n) { if (n > f.size() || pos > f.size() - n) throw std::runtime_error("Failure when loading image: truncated " + filename); };
    auto rd32 = [&]() { need(4); uint32_t v = f[pos] | (f[pos + 1] << 8) | (f[pos + 2] << 16) | ((uint32_t)f[pos + 3] << 24); pos += 4; return v; };
    auto rdstr = [&]() { std::string s; for (;;) { need(1); char c = (char)f[pos++]; if (!c) break; s.push_back(c); } return s; };
    if (rd32() != 20000630u) throw std::runtime_error("Failure when loading image: not an OpenEXR file: " + filename);
    uint32_t version = rd32();
    if (version & 0x1E00u) throw std::runtime_error("Unsupported image format: tiled / deep / multi-part OpenEXR: " + filename);
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1, line_order = 0;
    for (;;) {
        std::string name = rdstr();
        if (name.empty()) break;
        std::string type = rdstr();
        uint32_t n = rd32();
        need(n);
        const size_t vpos = pos;
        if (name == "channels") {
            size_t p = vpos;
            const size_t vend = vpos + n;
            while (p < vend && f[p]) {
                Chan c;
                while (p < vend && f[p]) c.name.push_back((char)f[p++]);
                p++;
                if (p + 16 > vend) throw std::runtime_error("Failure when loading image: bad OpenEXR channel list: " + filename);
                c.type = (int)(f[p] | (f[p + 1] << 8));
                if (c.type < 0 || c.type > 2) throw std::runtime_error("Failure when loading image: bad OpenEXR pixel type: " + filename);
                p += 4 + 4;                               // pixel type, pLinear + reserved
                uint32_t xs = f[p] | (f[p + 1] << 8), ys = f[p + 4] | (f[p + 5] << 8);
                p += 8;
                if (xs != 1 || ys != 1) throw std::runtime_error("Unsupported image format: subsampled OpenEXR channel: " + filename);
                chans.push_back(c);
            }
        } else if (name == "compression" && n >= 1) compression = f[vpos];
        else if (name == "dataWindow" && n >= 16) { pos = vpos; x0 = (int)rd32(); y0 = (int)rd32(); x1 = (int)rd32(); y1 = (int)rd32(); }
        else if (name == "lineOrder" && n >= 1) line_order = f[vpos];
        pos = vpos + n;
    }
    (void)line_order;   // blocks carry their own y coordinate
    const long long wl = (long long)x1 - x0 + 1, hl = (long long)y1 - y0 + 1;
    if (wl <= 0 || hl <= 0 || wl > 65536 || hl > 65536 || wl * hl > (1LL << 28) || chans.empty() || chans.size() > 64)
        throw std::runtime_error("Failure when loading image: bad OpenEXR header: " + filename);
    const int w = (int)wl, h = (int)hl;
    int lines_per_block;
    if (compression == 0 || compression == 1 || compression == 2) lines_per_block = 1;
    else if (compression == 3) lines_per_block = 16;
    else if (compression == 4) lines_per_block = 32;
    else throw std::runtime_error("Unsupported image format: OpenEXR compression " + std::to_string(compression) + " (NONE, RLE, ZIPS, ZIP, PIZ are read): " + filename);
    size_t bytes_per_px = 0;
    for (auto &c : chans) bytes_per_px += (c.type == 1) ? 2 : 4;
    int ir = -1, ig = -1, ib = -1;
    for (size_t i = 0; i < chans.size(); i++) { if (chans[i].name == "R") ir = (int)i; if (chans[i].name == "G") ig = (int)i; if (chans[i].name == "B") ib = (int)i; }
    if (chans.size() == 1) ir = ig = ib = 0;
    if (ir < 0 || ig < 0 || ib < 0) throw std::runtime_error("Failure when loading image: OpenEXR without R, G, B channels: " + filename);
    const int nblocks = (h + lines_per_block - 1) / lines_per_block;
    need((size_t)nblocks * 8);
    std::vector<uint64_t> offs((size_t)nblocks);
    for (int i = 0; i < nblocks; i++) { uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)f[pos + k] << (8 * k); pos += 8; offs[(size_t)i] = v; }
    rgb->assign((size_t)w * h * 3, 0.f);
    for (int blk = 0; blk < nblocks; blk++) {
        pos = (size_t)offs[(size_t)blk];
        const long long byl = (long long)(int)rd32() - y0;
        uint32_t len = rd32();
        need(len);
        const int by = (byl < 0 || byl >= h) ? -1 : (int)byl;
        if (by < 0 || by >= h) throw std::runtime_error("Failure when loading image: bad OpenEXR block: " + filename);
        const int lines = std::min(lines_per_block, h - by);
        std::vector<unsigned char> raw((size_t)lines * w * bytes_per_px);
        if (compression == 0) { if (len != raw.size()) throw std::runtime_error("Failure when loading image: bad OpenEXR block size: " + filename); std::memcpy(raw.data(), &f[pos], len); }
        else if (compression == 1) exr_rle_decode(&f[pos], len, raw);
        else if (compression == 4) {
            std::vector<int> words;
            for (auto &c : chans) words.push_back(c.type == 1 ? 1 : 2);
            exr_piz_decode(&f[pos], len, raw, words, w, lines);
        }
        else exr_zip_decode(&f[pos], len, raw);
        size_t p = 0;
        for (int ly = 0; ly < lines; ly++)
            for (size_t c = 0; c < chans.size(); c++) {
                for (int x = 0; x < w; x++) {
                    float v;
                    if (chans[c].type == 1) { v = half_to_float((uint16_t)(raw[p] | (raw[p + 1] << 8))); p += 2; }
                    else if (chans[c].type == 2) { std::memcpy(&v, &raw[p], 4); p += 4; }
                    else { uint32_t u; std::memcpy(&u, &raw[p], 4); v = (float)u; p += 4; }
                    float *px = &(*rgb)[((size_t)(by + ly) * w + x) * 3];
                    if ((int)c == ir) px[0] = v;
                    if ((int)c == ig) px[1] = v;
                    if ((int)c == ib) px[2] = v;
                }
            }
    }
    *width = w; *height = h;
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
    {
        std::string low = path;
        for (auto &ch : low) ch = (char)std::tolower((unsigned char)ch);
        if (ends_with(low, ".exr")) {
            // imread3: (R, G, B); imread1: (R + G + B) / 3 evaluated in fp32 like the reference (src/image.cpp:63-65)
            std::vector<float> rgb;
            read_exr_rgb(path, width, height, &rgb);
            const size_t n = (size_t)*width * *height;
            texels->resize(n * channels);
            for (size_t i = 0; i < n; i++) {
                if (channels == 3) for (int c = 0; c < 3; c++) (*texels)[3 * i + c] = (double)rgb[3 * i + c];
                else (*texels)[i] = (double)((rgb[3 * i] + rgb[3 * i + 1] + rgb[3 * i + 2]) / 3);
            }
            return;
        }
        const bool is_png = ends_with(low, ".png");
        if (ends_with(low, ".jpg") || ends_with(low, ".jpeg") || is_png) {
            // decoded natively; 8-bit -> linear exactly as stbi_loadf widens it: (float)(pow(v / 255.0f, 2.2f) * 1.0f),
            // the float overload of pow as image.cpp (C++) resolves it (src/3rdparty/stb_image.h:1849)
            std::ifstream jf(path, std::ios::binary);
            if (!jf) throw std::runtime_error("Failure when loading image: " + path);
            std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(jf)), std::istreambuf_iterator<char>());
            std::vector<uint8_t> px;
            try {
                if (is_png) decode_png(bytes.data(), bytes.size(), channels, width, height, &px);
                else decode_jpeg(bytes.data(), bytes.size(), channels, width, height, &px);
            } catch (const std::exception &e) {
                throw std::runtime_error("Failure when loading image: " + path + " (" + e.what() + ")");
            }
            float lut[256];
            for (int v = 0; v < 256; v++) lut[v] = (float)(std::pow(v / 255.0f, 2.2f) * 1.0f);
            texels->resize(px.size());
            for (size_t i = 0; i < px.size(); i++) (*texels)[i] = (double)lut[px[i]];
            return;
        }
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
    // sizes are checked against what the file can hold before anything is allocated from them
    if (hdr[0] > 65536 || hdr[1] > 65536) throw std::runtime_error("Failure when loading image: implausible extent in " + raw);
    const std::streamoff body = ifs.tellg();
    ifs.seekg(0, std::ios::end);
    const uint64_t remaining = (uint64_t)(ifs.tellg() - body);
    ifs.seekg(body);
    *width = hdr[0]; *height = hdr[1];
    int fc = hdr[2];
    const uint64_t ntex = (uint64_t)hdr[0] * (uint64_t)hdr[1] * (uint64_t)fc;
    if (v1 && ntex * sizeof(float) > remaining) throw std::runtime_error("Failure when loading image: truncated " + raw);
    // v2: zlib expands at most ~1032:1, so a body this short cannot hold that many texels
    if (v2 && (remaining < 4 || ntex > (remaining - 4) * 1040 + 64)) throw std::runtime_error("Failure when loading image: truncated " + raw);
    std::vector<float> data((size_t)ntex);
    if (v1) {
        ifs.read((char *)data.data(), (std::streamsize)(data.size() * sizeof(float)));
        if (!ifs) throw std::runtime_error("Failure when loading image: truncated " + raw);
    } else {
        // 8-bit texels, zlib-compressed; widened like stbi_loadf: (float)pow(v/255.0f, 2.2f) (stb_image.h:1849).
        // A 1-channel request is reduced in 8 bits first, as stb does: y = (77 r + 150 g + 29 b) >> 8 (stb_image.h:1708).
        uint32_t zlen = 0;
        ifs.read((char *)&zlen, 4);
        if (!ifs || (uint64_t)zlen > remaining - 4) throw std::runtime_error("Failure when loading image: truncated " + raw);
        std::vector<unsigned char> z(zlen), u8((size_t)ntex);
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
