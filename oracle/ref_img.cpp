// ref_img.cpp — reference-side image I/O known answers (TEST INFRASTRUCTURE; built only where /root/reference exists).
// Links the reference's own src/image.cpp (stb_image + tinyexr + miniz as vendored there) and reports what
// imread3()/imread1() (src/image.cpp:26-133) return and what imwrite() (src/image.cpp:135-173) writes, so that the
// build's own JPEG decoder and EXR writer/reader can be pinned against the reference's behaviour.
//
//   ref_img hash3 <image file>...      -> JSON: per file width, height, FNV-1a-64 of the fp32 texel bits, first texels
//   ref_img hash1 <image file>...      -> same through imread1()
//   ref_img readexr <file.exr>         -> JSON: width, height, FNV-1a-64 of the fp32 RGB values LoadEXR returns
//   ref_img writeexr <out.exr> <w> <h> -> writes the deterministic test image through the reference's imwrite()
//   ref_img writepfm <out.pfm> <w> <h>
#include "image.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

static uint64_t fnv1a(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static uint32_t crc32_update(uint32_t crc, const void *p, size_t n) {      // zlib's CRC-32 (reflected 0xEDB88320)
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    const unsigned char *b = (const unsigned char *)p;
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ b[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

// deterministic test picture shared with tests/test_image_io.py (values chosen to exercise half rounding:
// subnormals, exact ties, large values, negatives)
static Vector3 test_pixel(int x, int y, int w, int h) {
    double u = (x + 0.5) / w, v = (y + 0.5) / h;
    double r = u * u * 3.0 - 0.25, g = (v - 0.5) * 1e-5, b = 100.0 * u * v + 1.0 / 1024.0 + (x == 1 && y == 1 ? 70000.0 : 0.0);
    return Vector3{r, g, b};
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_img hash3|hash1|readexr|writeexr|writepfm ...\n"); return 2; }
    std::string mode = argv[1];
    if (mode == "hash3" || mode == "hash1") {
        printf("{");
        for (int i = 2; i < argc; i++) {
            std::string name = argv[i];
            size_t slash = name.find_last_of('/');
            std::string base = slash == std::string::npos ? name : name.substr(slash + 1);
            uint64_t h = 1469598103934665603ull;
            uint32_t crc = 0;
            int w, hgt;
            double sum = 0;
            float first[6] = {0, 0, 0, 0, 0, 0};
            if (mode == "hash3") {
                Image3 im = imread3(name);
                w = im.width; hgt = im.height;
                for (int k = 0; k < w * hgt; k++) {
                    float t[3] = {(float)im(k)[0], (float)im(k)[1], (float)im(k)[2]};
                    h = fnv1a(t, sizeof(t), h);
                    crc = crc32_update(crc, t, sizeof(t));
                    sum += t[0] + t[1] + t[2];
                    if (k < 2) { first[3 * k] = t[0]; first[3 * k + 1] = t[1]; first[3 * k + 2] = t[2]; }
                }
            } else {
                Image1 im = imread1(name);
                w = im.width; hgt = im.height;
                for (int k = 0; k < w * hgt; k++) {
                    float t = (float)im(k);
                    h = fnv1a(&t, sizeof(t), h);
                    crc = crc32_update(crc, &t, sizeof(t));
                    sum += t;
                    if (k < 6) first[k] = t;
                }
            }
            printf("%s\"%s\":{\"width\":%d,\"height\":%d,\"fnv1a64\":\"%016llx\",\"crc32\":%u,\"sum\":%.17g,\"first\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g]}",
                   i > 2 ? "," : "", base.c_str(), w, hgt, (unsigned long long)h, crc, sum, first[0], first[1], first[2], first[3], first[4], first[5]);
        }
        printf("}\n");
        return 0;
    }
    if (mode == "readexr") {
        Image3 im = imread3(argv[2]);
        uint32_t crc = 0;
        for (int k = 0; k < im.width * im.height; k++) {
            float t[3] = {(float)im(k)[0], (float)im(k)[1], (float)im(k)[2]};
            crc = crc32_update(crc, t, sizeof(t));
        }
        printf("{\"width\":%d,\"height\":%d,\"crc32\":%u}\n", im.width, im.height, crc);
        return 0;
    }
    if ((mode == "writeexr" || mode == "writepfm") && argc >= 5) {
        int w = atoi(argv[3]), h = atoi(argv[4]);
        Image3 im(w, h);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) im(x, y) = test_pixel(x, y, w, h);
        imwrite(argv[2], im);
        return 0;
    }
    fprintf(stderr, "bad arguments\n");
    return 2;
}
