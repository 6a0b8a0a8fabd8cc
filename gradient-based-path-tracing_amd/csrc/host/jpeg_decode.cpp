// jpeg_decode.cpp — baseline and progressive (Huffman, 8-bit) JPEG decoder for <texture type="bitmap"> inputs.
//
// The reference reads bitmaps through stb_image v2.27 (stbi_loadf in imread1/imread3, src/image.cpp:26-108; the
// library is vendored at src/3rdparty/stb_image.h). Texels are data on the hot path (Q6 of SURVEY §8), so this decoder
// has to return the very 8-bit samples stb returns. The entropy-coded stream decodes to the same coefficients in any
// conforming decoder; what defines the output is the arithmetic after it, restated here from stb_image:
//   * dequantised coefficients are truncated to int16                                   (stb_image.h:2199,2227)
//   * inverse DCT: the jidctint-ISLOW-derived integer transform with 12-bit constants, column pass keeping 2 extra
//     bits (+512 >> 10), row pass rounding with 65536 + (128 << 17) >> 17                 (:2392-2489)
//   * chroma upsampling: 3:1 triangle filters, (3a+b+2)>>2 vertically / horizontally and
//     (3*t0+t1+8)>>4 for 2x2                                                            (:3411-3474)
//   * YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits first     (:3604-3630)
//   * a 1-channel request of a YCbCr file returns the Y plane itself                      (:3826-3829)
//   * progressive files (SOF2; T.81 Annex G: spectral selection + successive approximation) accumulate their
//     coefficients over the scans as int16, first DC scan zeroing the block; they are dequantised (product truncated to
//     int16, with the tables in effect at the end of the file) and transformed once, after the last scan   (:2113-2290, :3092-3113)
// stb's SIMD kernels are written to be bit-identical to these scalar forms (its own comments at :2492, :3604).
// Pinned by tests/golden/ref_textures.json, made with the reference's own image.cpp (oracle/ref_img.cpp).
#include "jpeg_decode.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace gdpt {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    // canonical code description (ITU T.81 Annex C): codes of length l occupy [first[l], first[l] + count[l])
    int count[17] = {0};
    int first_code[17] = {0};
    int first_index[17] = {0};
    uint8_t symbols[256] = {0};
    bool defined = false;
    void build(const int counts[16], const uint8_t *syms, int n) {
        int code = 0, idx = 0;
        for (int l = 1; l <= 16; l++) {
            count[l] = counts[l - 1];
            first_code[l] = code;
            first_index[l] = idx;
            code = (code + count[l]) << 1;
            idx += count[l];
        }
        if (idx != n || n > 256) throw std::runtime_error("bad Huffman table");
        std::memcpy(symbols, syms, (size_t)n);
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;    // sample extent, padded plane extent
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;          // progressive files: 64 coefficients per block of the padded plane, block-row-major
    int coeff_w = 0;                     // blocks per row of `coeff`
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int bits = 0;
    int marker = 0;        // marker met inside the entropy-coded segment (0 = none)
    void reset() { acc = 0; bits = 0; marker = 0; }
    void fill() {
        while (bits <= 24) {
            uint32_t b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p++ : 0xD9;
                    while (c == 0xFF && p < end) c = *p++;          // fill bytes
                    if (c != 0) { marker = c; b = 0; }
                }
            }
            acc |= b << (24 - bits);
            bits += 8;
        }
    }
    int get_bit() {
        if (bits < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1; bits--;
        return b;
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n; bits -= n;
        return v;
    }
    int decode(const HuffTable &t) {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | get_bit();
            int off = code - t.first_code[l];
            if (off >= 0 && off < t.count[l]) return t.symbols[t.first_index[l] + off];
        }
        throw std::runtime_error("bad Huffman code");
    }
    // T.81 F.2.2.1 RECEIVE + EXTEND
    int receive_extend(int n) {
        if (n == 0) return 0;
        int v = get_bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
};

inline uint8_t clamp8(long long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// 12-bit constants exactly as (int)(c * 4096 + 0.5) evaluates in stb (c is a float literal there)
inline int fx(float c) { return (int)(c * 4096 + 0.5); }

// 64-bit intermediates: identical to stb's 32-bit arithmetic for every valid stream (whose values stay far inside 32
// bits) and free of signed overflow on corrupt ones.
typedef long long I64;
// One 8-point inverse DCT pass. Source of the arithmetic: the Loeffler-Ligtenberg-Moschytz factorisation with 12-bit
// fixed-point constants as the Independent JPEG Group's "slow integer" IDCT (jidctint.c) states it and as stb_image
// evaluates it; bit-exact texels (tests/golden/ref_images.json holds the reference decoder's CRCs) admit no other
// operation order, so the sums below keep that order. Output: even-part sums e[0..3] and odd-part sums o[0..3]; sample
// k of the pass is e[k] + o[3-k] (k < 4) and e[7-k] - o[k-4] (k >= 4) after the caller's rounding shift.
struct Idct1D { I64 x0, x1, x2, x3, t0, t1, t2, t3; };     // x = even part, t = odd part
inline Idct1D idct_1d(I64 s0, I64 s1, I64 s2, I64 s3, I64 s4, I64 s5, I64 s6, I64 s7) {
    Idct1D r;
    // even part: rotation of (s2, s6) by 6*pi/16, butterfly of (s0, s4)
    const I64 rot = (s2 + s6) * fx(0.5411961f);
    const I64 even_b = rot + s6 * fx(-1.847759065f);
    const I64 even_a = rot + s2 * fx(0.765366865f);
    const I64 sum04 = (s0 + s4) * 4096, dif04 = (s0 - s4) * 4096;
    r.x0 = sum04 + even_a; r.x3 = sum04 - even_a; r.x1 = dif04 + even_b; r.x2 = dif04 - even_b;
    // odd part: four inputs, three shared rotations (the "z" terms of the factorisation)
    const I64 z73 = s7 + s3, z51 = s5 + s1, z71 = s7 + s1, z53 = s5 + s3;
    const I64 zall = (z73 + z51) * fx(1.175875602f);
    const I64 q71 = zall + z71 * fx(-0.899976223f);
    const I64 q53 = zall + z53 * fx(-2.562915447f);
    const I64 q73 = z73 * fx(-1.961570560f);
    const I64 q51 = z51 * fx(-0.390180644f);
    r.t3 = s1 * fx(1.501321110f) + q71 + q51;
    r.t2 = s3 * fx(3.072711026f) + q53 + q73;
    r.t1 = s5 * fx(2.053119869f) + q53 + q51;
    r.t0 = s7 * fx(0.298631336f) + q71 + q73;
    return r;
}

void idct_block(uint8_t *out, int stride, const int16_t d[64]) {
    I64 val[64];
    for (int i = 0; i < 8; i++) {                       // columns
        const int16_t *c = d + i;
        I64 *v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            I64 dc = c[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            Idct1D r = idct_1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
            r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
            v[0] = (r.x0 + r.t3) >> 10; v[56] = (r.x0 - r.t3) >> 10;
            v[8] = (r.x1 + r.t2) >> 10; v[48] = (r.x1 - r.t2) >> 10;
            v[16] = (r.x2 + r.t1) >> 10; v[40] = (r.x2 - r.t1) >> 10;
            v[24] = (r.x3 + r.t0) >> 10; v[32] = (r.x3 - r.t0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {                       // rows
        const I64 *v = val + 8 * i;
        uint8_t *o = out + (size_t)i * stride;
        Idct1D r = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const I64 bias = 65536 + (128 << 17);
        r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
        o[0] = clamp8((r.x0 + r.t3) >> 17); o[7] = clamp8((r.x0 - r.t3) >> 17);
        o[1] = clamp8((r.x1 + r.t2) >> 17); o[6] = clamp8((r.x1 - r.t2) >> 17);
        o[2] = clamp8((r.x2 + r.t1) >> 17); o[5] = clamp8((r.x2 - r.t1) >> 17);
        o[3] = clamp8((r.x3 + r.t0) >> 17); o[4] = clamp8((r.x3 - r.t0) >> 17);
    }
}

struct Decoder {
    const uint8_t *data;
    size_t size, pos = 0;
    int width = 0, height = 0, ncomp = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    bool jfif = false;
    int adobe_transform = -1;
    int rgb_ids = 0;
    bool have_frame = false, scanned = false;
    bool progressive = false;
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;     // progressive scan parameters
    uint16_t quant[4][64];
    HuffTable dc_tab[4], ac_tab[4];
    Component comp[4];

    int u8() { if (pos >= size) throw std::runtime_error("truncated JPEG"); return data[pos++]; }
    int u16() { int a = u8(); return (a << 8) | u8(); }

    int next_marker() {
        int c = u8();
        if (c != 0xFF) throw std::runtime_error("expected marker");
        while (c == 0xFF) c = u8();
        return c;
    }

    void read_dqt() {
        int L = u16() - 2;
        while (L > 0) {
            int q = u8(), p = q >> 4, t = q & 15;
            if (p > 1 || t > 3) throw std::runtime_error("bad DQT");
            for (int i = 0; i < 64; i++) quant[t][kZigzag[i]] = (uint16_t)(p ? u16() : u8());
            L -= p ? 129 : 65;
        }
        if (L != 0) throw std::runtime_error("bad DQT length");
    }
    void read_dht() {
        int L = u16() - 2;
        while (L > 0) {
            int q = u8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) throw std::runtime_error("bad DHT");
            int counts[16], n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = u8(); n += counts[i]; }
            if (n > 256) throw std::runtime_error("bad DHT");
            uint8_t syms[256];
            for (int i = 0; i < n; i++) syms[i] = (uint8_t)u8();
            (tc ? ac_tab[th] : dc_tab[th]).build(counts, syms, n);
            L -= 17 + n;
        }
        if (L != 0) throw std::runtime_error("bad DHT length");
    }
    void read_app(int m) {
        int L = u16();
        if (L < 2) throw std::runtime_error("bad APP length");
        L -= 2;
        if (m == 0xE0 && L >= 5) {
            static const char tag[5] = {'J', 'F', 'I', 'F', 0};
            bool ok = true;
            for (int i = 0; i < 5; i++) if (u8() != (uint8_t)tag[i]) ok = false;
            L -= 5;
            if (ok) jfif = true;
        } else if (m == 0xEE && L >= 12) {
            static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
            bool ok = true;
            for (int i = 0; i < 6; i++) if (u8() != (uint8_t)tag[i]) ok = false;
            L -= 6;
            if (ok) { u8(); u16(); u16(); adobe_transform = u8(); L -= 6; }
        }
        if (pos + (size_t)L > size) throw std::runtime_error("truncated JPEG");
        pos += (size_t)L;
    }
    void read_sof(int m) {
        progressive = (m == 0xC2);
        int Lf = u16();
        if (u8() != 8) throw std::runtime_error("JPEG: only 8-bit samples");
        height = u16(); width = u16();
        if (height == 0 || width == 0) throw std::runtime_error("JPEG: zero extent");
        ncomp = u8();
        if (ncomp != 1 && ncomp != 3) throw std::runtime_error("JPEG: 1 or 3 components expected");
        if (Lf != 8 + 3 * ncomp) throw std::runtime_error("bad SOF length");
        static const uint8_t rgb[3] = {'R', 'G', 'B'};
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.id = u8();
            if (ncomp == 3 && c.id == rgb[i]) rgb_ids++;
            int q = u8();
            c.h = q >> 4; c.v = q & 15; c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw std::runtime_error("bad SOF component");
            h_max = c.h > h_max ? c.h : h_max; v_max = c.v > v_max ? c.v : v_max;
        }
        for (int i = 0; i < ncomp; i++) if (h_max % comp[i].h || v_max % comp[i].v) throw std::runtime_error("fractional sampling ratio");
        mcu_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcu_y = (height + 8 * v_max - 1) / (8 * v_max);
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) { c.coeff_w = c.w2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
        }
        have_frame = true;
    }

    void decode_block(BitReader &br, Component &c, int16_t blk[64]) {
        const HuffTable &hd = dc_tab[c.td], &ha = ac_tab[c.ta];
        if (!hd.defined || !ha.defined) throw std::runtime_error("missing Huffman table");
        const uint16_t *dq = quant[c.tq];
        std::memset(blk, 0, 64 * sizeof(int16_t));
        int t = br.decode(hd);
        if (t > 15) throw std::runtime_error("bad DC size");
        int dc = c.dc_pred + br.receive_extend(t);
        c.dc_pred = dc;
        blk[0] = (int16_t)(dc * dq[0]);
        for (int k = 1; k < 64;) {
            int rs = br.decode(ha), s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) throw std::runtime_error("bad AC run");
                int z = kZigzag[k++];
                blk[z] = (int16_t)(br.receive_extend(s) * dq[z]);
            }
        }
    }

    // ---- progressive scans (T.81 G.1.2): coefficients accumulate in Component::coeff, un-dequantised ----
    void prog_dc(BitReader &br, Component &c, int16_t *blk) {
        if (spec_end != 0) throw std::runtime_error("progressive JPEG: DC and AC in one scan");
        if (succ_high == 0) {                           // first DC scan: the block starts here
            std::memset(blk, 0, 64 * sizeof(int16_t));
            const HuffTable &hd = dc_tab[c.td];
            if (!hd.defined) throw std::runtime_error("missing Huffman table");
            int t = br.decode(hd);
            if (t > 15) throw std::runtime_error("bad DC size");
            int dc = c.dc_pred + br.receive_extend(t);
            c.dc_pred = dc;
            blk[0] = (int16_t)(dc * (1 << succ_low));
        } else if (br.get_bit()) blk[0] = (int16_t)(blk[0] + (int16_t)(1 << succ_low));      // refinement: one more bit
    }
    void prog_ac(BitReader &br, Component &c, int16_t *blk) {
        if (spec_start == 0) throw std::runtime_error("progressive JPEG: DC and AC in one scan");
        const HuffTable &ha = ac_tab[c.ta];
        if (!ha.defined) throw std::runtime_error("missing Huffman table");
        if (succ_high == 0) {                           // first pass over this band
            if (eob_run) { --eob_run; return; }
            int k = spec_start;
            do {
                int rs = br.decode(ha), s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {                       // end of band for 2^r (+ extra bits) blocks, this one included
                        eob_run = 1 << r;
                        if (r) eob_run += br.get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) throw std::runtime_error("bad AC run");
                    int z = kZigzag[k++];
                    blk[z] = (int16_t)(br.receive_extend(s) * (1 << succ_low));
                }
            } while (k <= spec_end);
        } else {                                        // refinement pass: one more bit for known coefficients, new +-1s
            const int16_t bit = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t *p) { if (br.get_bit() && (*p & bit) == 0) *p = (int16_t)(*p > 0 ? *p + bit : *p - bit); };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; k++) { int16_t *p = &blk[kZigzag[k]]; if (*p != 0) refine(p); }
            } else {
                int k = spec_start;
                do {
                    int rs = br.decode(ha), s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += br.get_bits(r);
                            r = 64;                     // run to the end of the band, refining what is there
                        }                               // r == 15: sixteen zeros = a run of 15 and a zero "new" value
                    } else {
                        if (s != 1) throw std::runtime_error("bad AC refinement");
                        s = br.get_bit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        int16_t *p = &blk[kZigzag[k++]];
                        if (*p != 0) refine(p);
                        else {
                            if (r == 0) { *p = (int16_t)s; break; }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
    }
    // after the last scan: dequantise (truncated to int16, like the sequential path) and transform every block
    void finish_progressive() {
        for (int n = 0; n < ncomp; n++) {
            Component &c = comp[n];
            const uint16_t *dq = quant[c.tq];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    int16_t *blk = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int k = 0; k < 64; k++) blk[k] = (int16_t)(blk[k] * dq[k]);
                    idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, blk);
                }
        }
    }

    void read_scan() {
        if (!have_frame) throw std::runtime_error("SOS before SOF");
        int Ls = u16(), n = u8();
        if (n < 1 || n > ncomp || Ls != 6 + 2 * n) throw std::runtime_error("bad SOS");
        int order[4];
        for (int i = 0; i < n; i++) {
            int id = u8(), q = u8(), which = -1;
            for (int k = 0; k < ncomp; k++) if (comp[k].id == id) which = k;
            if (which < 0) throw std::runtime_error("bad SOS component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) throw std::runtime_error("bad SOS table");
            order[i] = which;
        }
        spec_start = u8(); spec_end = u8();
        { int ahl = u8(); succ_high = ahl >> 4; succ_low = ahl & 15; }
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) throw std::runtime_error("bad SOS (progressive)");
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) throw std::runtime_error("bad SOS (not sequential)");
            spec_end = 63;
        }
        BitReader br{data + pos, data + size};
        for (int k = 0; k < ncomp; k++) comp[k].dc_pred = 0;
        eob_run = 0;
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        int16_t blk[64];
        auto restart_check = [&]() {
            if (--todo > 0) return true;
            if (br.bits < 24) br.fill();
            if (!(br.marker >= 0xD0 && br.marker <= 0xD7)) return false;     // no restart marker: the scan is over
            br.reset();
            for (int k = 0; k < ncomp; k++) comp[k].dc_pred = 0;
            eob_run = 0;
            todo = restart_interval ? restart_interval : 0x7fffffff;
            return true;
        };
        auto block_at = [&](Component &c, int bx, int by) { return c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.coeff_w); };
        if (n == 1) {                                   // non-interleaved: the component's own block grid
            Component &c = comp[order[0]];
            int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            bool go = true;
            for (int j = 0; j < bh && go; j++)
                for (int i = 0; i < bw && go; i++) {
                    if (progressive) { if (spec_start == 0) prog_dc(br, c, block_at(c, i, j)); else prog_ac(br, c, block_at(c, i, j)); }
                    else {
                        decode_block(br, c, blk);
                        idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, blk);
                    }
                    go = restart_check();
                }
        } else {                                        // interleaved MCUs
            bool go = true;
            for (int j = 0; j < mcu_y && go; j++)
                for (int i = 0; i < mcu_x && go; i++) {
                    for (int k = 0; k < n; k++) {
                        Component &c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                                if (progressive) { prog_dc(br, c, block_at(c, x2 / 8, y2 / 8)); continue; }     // (interleaved scans carry DC only)
                                decode_block(br, c, blk);
                                idct_block(c.plane.data() + (size_t)c.w2 * y2 + x2, c.w2, blk);
                            }
                    }
                    go = restart_check();
                }
        }
        // continue parsing after the entropy-coded data: at the marker the bit reader ran into, or scan forward
        if (br.marker) {
            // br.p points just past the marker code
            pos = (size_t)(br.p - data) - 2;
        } else {
            pos = (size_t)(br.p - data);
            while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] != 0 && data[pos + 1] != 0xFF && !(data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7))) pos++;
        }
        scanned = true;
    }

    void parse() {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw std::runtime_error("not a JPEG (no SOI)");
        pos = 2;
        for (;;) {
            if (pos >= size) break;                     // tolerate a missing EOI once a scan was decoded
            int m = next_marker();
            if (m == 0xD9) break;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) read_sof(m);
            else if (m == 0xC4) read_dht();
            else if (m == 0xDB) read_dqt();
            else if (m == 0xDD) { if (u16() != 4) throw std::runtime_error("bad DRI"); restart_interval = u16(); }
            else if (m == 0xDA) read_scan();
            else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) read_app(m);
            else if (m >= 0xD0 && m <= 0xD7) continue;
            else {                                      // other segments carry a length
                int L = u16();
                if (L < 2 || pos + (size_t)(L - 2) > size) throw std::runtime_error("bad JPEG segment");
                pos += (size_t)(L - 2);
            }
        }
        if (!scanned) throw std::runtime_error("JPEG without image data");
        if (progressive) finish_progressive();
    }
};

// ---- upsampling of one output row (stb_image.h:3403-3474 + the generic nearest-neighbour case) ----
const uint8_t *resample_row(std::vector<uint8_t> &buf, const uint8_t *near_, const uint8_t *far_, int w, int hs, int vs) {
    uint8_t *out = buf.data();
    if (hs == 1 && vs == 1) return near_;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        const uint8_t *in = near_;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            int n = 3 * in[i] + 2;
            out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return out; }
        int t1 = 3 * near_[0] + far_[0], t0;
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            t0 = t1;
            t1 = 3 * near_[i] + far_[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near_[i];
    return out;
}

inline int f2fixed(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

} // namespace

void decode_jpeg(const uint8_t *bytes, size_t size, int req_comp, int *width, int *height, std::vector<uint8_t> *out) {
    if (req_comp != 1 && req_comp != 3) throw std::runtime_error("decode_jpeg: 1 or 3 output channels");
    Decoder d{bytes, size};
    d.parse();
    const int W = d.width, H = d.height;
    const bool is_rgb = d.ncomp == 3 && (d.rgb_ids == 3 || (d.adobe_transform == 0 && !d.jfif));
    const int decode_n = (d.ncomp == 3 && req_comp < 3 && !is_rgb) ? 1 : d.ncomp;
    struct Res { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
    Res res[3];
    for (int k = 0; k < decode_n; k++) {
        Res &r = res[k];
        r.hs = d.h_max / d.comp[k].h; r.vs = d.v_max / d.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (W + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = d.comp[k].plane.data();
        r.buf.assign((size_t)W + 8, 0);
    }
    out->assign((size_t)W * H * req_comp, 0);
    const uint8_t *row[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < H; j++) {
        uint8_t *o = out->data() + (size_t)req_comp * W * j;
        for (int k = 0; k < decode_n; k++) {
            Res &r = res[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            row[k] = resample_row(r.buf, y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        if (req_comp == 3) {
            if (d.ncomp == 3 && !is_rgb) {
                for (int i = 0; i < W; i++) {
                    int y_fixed = (row[0][i] << 20) + (1 << 19);
                    int cr = row[2][i] - 128, cb = row[1][i] - 128;
                    int r = y_fixed + cr * f2fixed(1.40200f);
                    int g = y_fixed + (cr * -f2fixed(0.71414f)) + ((cb * -f2fixed(0.34414f)) & 0xffff0000);
                    int b = y_fixed + cb * f2fixed(1.77200f);
                    o[3 * i] = clamp8(r >> 20); o[3 * i + 1] = clamp8(g >> 20); o[3 * i + 2] = clamp8(b >> 20);
                }
            } else if (d.ncomp == 3) {
                for (int i = 0; i < W; i++) { o[3 * i] = row[0][i]; o[3 * i + 1] = row[1][i]; o[3 * i + 2] = row[2][i]; }
            } else {
                for (int i = 0; i < W; i++) o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = row[0][i];
            }
        } else {
            if (is_rgb) for (int i = 0; i < W; i++) o[i] = (uint8_t)(((row[0][i] * 77) + (row[1][i] * 150) + (29 * row[2][i])) >> 8);
            else for (int i = 0; i < W; i++) o[i] = row[0][i];
        }
    }
    *width = W; *height = H;
}

} // namespace gdpt
