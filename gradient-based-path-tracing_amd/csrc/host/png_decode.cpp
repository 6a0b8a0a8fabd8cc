// png_decode.cpp — PNG decoder for <texture type="bitmap"> inputs whose 8-bit output equals what the reference gets.
//
// The reference reads bitmaps through stb_image v2.27 (stbi_loadf in imread1 / imread3, src/image.cpp:26-108, with 1 or 3
// requested channels). The PNG format itself (RFC 2083 / ISO 15948: chunks, zlib stream, the five scanline filters, Adam7)
// leaves no freedom in the decoded samples; what defines the texels is what stb_image does AFTER decoding, restated here:
//   * samples of 1, 2 and 4 bits are widened by 0xff, 0x55, 0x11 (grey) or kept as palette indices;
//   * a tRNS colour key adds an alpha channel (0 where the pixel equals the key, else opaque); a palette with tRNS expands
//     to RGBA; alpha is never applied to the colours, a request for 1 or 3 channels just drops it;
//   * 16-bit files are converted to the requested channel count in 16 bits first and then keep the HIGH byte;
//   * RGB -> grey is (77 r + 150 g + 29 b) >> 8 in integers; grey -> RGB replicates;
//   * gAMA, cHRM, sRGB, iCCP and every other ancillary chunk are ignored, chunk CRCs and the Adler-32 are not what stb
//     checks either (zlib's inflate here does verify the Adler-32: a corrupt stream is an error, not different texels).
// Pinned by tests/golden/ref_images.json ("png"): CRCs of the fp32 texels the reference's own image.cpp returns for the
// fixtures under tests/golden/images (oracle/ref_img.cpp).
#include "png_decode.h"

#include <zlib.h>

#include <cstring>
#include <stdexcept>
#include <string>

namespace gdpt {

namespace {

struct Reader {
    const uint8_t *p; size_t n, pos = 0;
    uint32_t u32() { if (pos + 4 > n) throw std::runtime_error("truncated PNG"); uint32_t v = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3]; pos += 4; return v; }
    uint8_t u8() { if (pos >= n) throw std::runtime_error("truncated PNG"); return p[pos++]; }
};

int paeth(int a, int b, int c) {
    const int pr = a + b - c, pa = pr > a ? pr - a : a - pr, pb = pr > b ? pr - b : b - pr, pc = pr > c ? pr - c : c - pr;
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// Unfilters `rows` scanlines of `stride` bytes each (each preceded by its filter byte) in place; bpp = bytes per complete pixel
// (1 for sub-byte depths). Returns a pointer behind the consumed input.
const uint8_t *unfilter(const uint8_t *src, const uint8_t *end, std::vector<uint8_t> &out, size_t stride, size_t rows, size_t bpp) {
    out.assign(stride * rows, 0);
    for (size_t y = 0; y < rows; y++) {
        if ((size_t)(end - src) < stride + 1) throw std::runtime_error("PNG: not enough pixel data");
        const int ft = *src++;
        if (ft > 4) throw std::runtime_error("PNG: invalid filter");
        uint8_t *cur = out.data() + y * stride;
        const uint8_t *up = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = src[i];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            cur[i] = (uint8_t)v;
        }
        src += stride;
    }
    return src;
}

} // namespace

void decode_png(const uint8_t *bytes, size_t size, int req_comp, int *width, int *height, std::vector<uint8_t> *out) {
    if (req_comp != 1 && req_comp != 3) throw std::runtime_error("decode_png: 1 or 3 output channels");
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (size < 8 || std::memcmp(bytes, sig, 8) != 0) throw std::runtime_error("not a PNG");
    Reader r{bytes, size, 8};
    uint32_t W = 0, H = 0;
    int depth = 0, color = 0, interlace = 0;
    bool have_ihdr = false, have_plte = false, has_trans = false;
    uint8_t palette[256][4];
    int pal_len = 0;
    uint16_t key[3] = {0, 0, 0};
    std::vector<uint8_t> z;
    for (bool first = true;; first = false) {
        const uint32_t len = r.u32(), type = r.u32();
        if (len > size - r.pos) throw std::runtime_error("truncated PNG");
        const uint8_t *body = bytes + r.pos;
        if (first && type != 0x49484452u) throw std::runtime_error("PNG: first chunk is not IHDR");
        if (type == 0x49484452u) {                              // IHDR
            if (have_ihdr || len != 13) throw std::runtime_error("PNG: bad IHDR");
            Reader h{body, 13};
            W = h.u32(); H = h.u32(); depth = h.u8(); color = h.u8();
            const int comp = h.u8(), filter = h.u8(); interlace = h.u8();
            if (W == 0 || H == 0 || W > (1u << 24) || H > (1u << 24)) throw std::runtime_error("PNG: bad extent");
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) throw std::runtime_error("PNG: 1/2/4/8/16-bit only");
            if (color > 6 || color == 1 || color == 5) throw std::runtime_error("PNG: bad colour type");
            if (color == 3 && depth == 16) throw std::runtime_error("PNG: bad colour type");
            if ((color == 2 || color == 4 || color == 6) && depth < 8) throw std::runtime_error("PNG: bad bit depth for the colour type");
            if (comp || filter || interlace > 1) throw std::runtime_error("PNG: bad compression / filter / interlace method");
            if ((uint64_t)W * H > (1ull << 28)) throw std::runtime_error("PNG: too large");
            have_ihdr = true;
        } else if (type == 0x504c5445u) {                       // PLTE
            if (len > 256 * 3 || len % 3) throw std::runtime_error("PNG: invalid PLTE");
            pal_len = (int)(len / 3);
            for (int i = 0; i < pal_len; i++) { palette[i][0] = body[3 * i]; palette[i][1] = body[3 * i + 1]; palette[i][2] = body[3 * i + 2]; palette[i][3] = 255; }
            have_plte = true;
        } else if (type == 0x74524e53u) {                       // tRNS
            if (!z.empty()) throw std::runtime_error("PNG: tRNS after IDAT");
            if (color == 3) {
                if (!have_plte || (int)len > pal_len) throw std::runtime_error("PNG: bad tRNS");
                for (uint32_t i = 0; i < len; i++) palette[i][3] = body[i];
                has_trans = true;
            } else {
                const int n = (color & 2) ? 3 : 1;
                if ((color & 4) || len != (uint32_t)n * 2) throw std::runtime_error("PNG: bad tRNS");
                for (int k = 0; k < n; k++) key[k] = (uint16_t)((body[2 * k] << 8) | body[2 * k + 1]);
                has_trans = true;
            }
        } else if (type == 0x49444154u) {                       // IDAT
            if (color == 3 && !have_plte) throw std::runtime_error("PNG: no PLTE");
            z.insert(z.end(), body, body + len);
        } else if (type == 0x49454e44u) break;                  // IEND
        else if (!(type & 0x20000000u)) throw std::runtime_error("PNG: unknown critical chunk");
        r.pos += len;
        r.u32();                                                // CRC (not verified, as in the reference's decoder)
    }
    if (!have_ihdr || z.empty()) throw std::runtime_error("PNG: no image data");

    const int img_n = ((color & 2) && color != 3 ? 3 : 1) + ((color & 4) ? 1 : 0);      // samples per pixel in the file
    const size_t bpp = depth < 8 ? 1 : (size_t)img_n * (depth / 8);
    auto row_bytes = [&](uint32_t w) { return ((size_t)w * img_n * depth + 7) >> 3; };
    // inflate: exactly the filtered scanlines of every pass
    static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t raw_len = 0;
    if (!interlace) raw_len = (row_bytes(W) + 1) * H;
    else for (int p = 0; p < 7; p++) {
        const uint32_t pw = (W - xo[p] + xs[p] - 1) / xs[p], ph = (H - yo[p] + ys[p] - 1) / ys[p];
        if (pw && ph) raw_len += (row_bytes(pw) + 1) * ph;
    }
    std::vector<uint8_t> raw(raw_len);
    {
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) throw std::runtime_error("PNG: zlib init failed");
        zs.next_in = z.data(); zs.avail_in = (uInt)z.size(); zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = raw.size() - zs.avail_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_BUF_ERROR && rc != Z_OK) || got < raw_len) throw std::runtime_error("PNG: corrupt or short zlib stream");
    }
    // samples of the whole image, one uint16 per sample (8-bit values for depth <= 8, palette indices for colour type 3)
    std::vector<uint16_t> smp((size_t)W * H * img_n);
    const int scale = (color == 0) ? (depth == 1 ? 0xff : depth == 2 ? 0x55 : depth == 4 ? 0x11 : 1) : 1;
    auto unpack = [&](const std::vector<uint8_t> &px, uint32_t pw, uint32_t ph, int x0, int y0, int dx, int dy) {
        const size_t stride = row_bytes(pw);
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t *row = px.data() + y * stride;
            for (uint32_t x = 0; x < pw; x++) {
                uint16_t *dst = &smp[(((size_t)y0 + (size_t)y * dy) * W + x0 + (size_t)x * dx) * img_n];
                if (depth == 16) for (int k = 0; k < img_n; k++) dst[k] = (uint16_t)((row[(x * img_n + k) * 2] << 8) | row[(x * img_n + k) * 2 + 1]);
                else if (depth == 8) for (int k = 0; k < img_n; k++) dst[k] = row[x * img_n + k];
                else {                                          // 1, 2, 4 bits: one sample per pixel, most significant bits first
                    const int per = 8 / depth, shift = (per - 1 - (int)(x % per)) * depth;
                    dst[0] = (uint16_t)(((row[x / per] >> shift) & ((1 << depth) - 1)) * scale);
                }
            }
        }
    };
    {
        const uint8_t *src = raw.data(), *end = raw.data() + raw.size();
        std::vector<uint8_t> px;
        if (!interlace) { src = unfilter(src, end, px, row_bytes(W), H, bpp); unpack(px, W, H, 0, 0, 1, 1); }
        else for (int p = 0; p < 7; p++) {
            const uint32_t pw = (W - xo[p] + xs[p] - 1) / xs[p], ph = (H - yo[p] + ys[p] - 1) / ys[p];
            if (!pw || !ph) continue;
            src = unfilter(src, end, px, row_bytes(pw), ph, bpp);
            unpack(px, pw, ph, xo[p], yo[p], xs[p], ys[p]);
        }
    }
    // to the requested channel count (alpha, from the file or from a colour key, is dropped: never applied)
    const size_t npx = (size_t)W * H;
    out->assign(npx * req_comp, 0);
    const bool deep = depth == 16;
    auto to8 = [&](unsigned v) { return (uint8_t)(deep ? (v >> 8) & 0xff : v); };
    for (size_t i = 0; i < npx; i++) {
        unsigned rgb[3];
        int nc;                                                 // colour channels of the source pixel: 1 or 3
        if (color == 3) {
            const unsigned idx = smp[i];
            // an index past the palette reads the reference decoder's zero-initialised entry... it has none: reject
            if ((int)idx >= pal_len) { if (idx >= 256) throw std::runtime_error("PNG: bad palette index"); rgb[0] = rgb[1] = rgb[2] = 0; }
            else { rgb[0] = palette[idx][0]; rgb[1] = palette[idx][1]; rgb[2] = palette[idx][2]; }
            nc = 3;
        } else if (color & 2) { rgb[0] = smp[i * img_n]; rgb[1] = smp[i * img_n + 1]; rgb[2] = smp[i * img_n + 2]; nc = 3; }
        else { rgb[0] = smp[i * img_n]; nc = 1; }
        uint8_t *o = out->data() + i * req_comp;
        if (req_comp == 3) {
            if (nc == 3) { o[0] = to8(rgb[0]); o[1] = to8(rgb[1]); o[2] = to8(rgb[2]); }
            else o[0] = o[1] = o[2] = to8(rgb[0]);
        } else {
            if (nc == 3) o[0] = to8(((rgb[0] * 77) + (rgb[1] * 150) + (29 * rgb[2])) >> 8);      // in 16 bits for deep files, then the high byte
            else o[0] = to8(rgb[0]);
        }
    }
    (void)has_trans; (void)key;          // the colour key only ever produces the alpha channel, which no caller asks for
    *width = (int)W; *height = (int)H;
}

} // namespace gdpt
