// device_math.h — fp64 vector math, PCG32 and frames for the HIP kernels (gfx950).
// Semantics follow the reference's src/vector.h, src/frame.h, src/pcg.h (cited per function).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GD __device__ __forceinline__

namespace gd {

constexpr double kPi = 3.14159265358979323846;    // c_PI, src/lajolla.h:25
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kInvPi = 1.0 / kPi;

struct D2 { double x, y; };
struct D3 { double x, y, z; };

GD D3 mk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
GD D3 splat(double v) { return mk(v, v, v); }
GD D3 operator+(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
GD D3 operator-(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
GD D3 operator-(D3 a) { return mk(-a.x, -a.y, -a.z); }
GD D3 operator*(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
GD D3 operator*(double s, D3 a) { return mk(a.x * s, a.y * s, a.z * s); }
GD D3 operator*(D3 a, D3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
GD D3 operator/(D3 a, double s) { double inv = 1.0 / s; return mk(a.x * inv, a.y * inv, a.z * inv); } // src/vector.h:194-197
GD double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GD D3 cross(D3 a, D3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GD double length(D3 a) { return sqrt(dot(a, a)); }
GD D3 normalize(D3 a) { double l = length(a); if (l <= 0) return mk(0, 0, 0); return a / l; } // src/vector.h:250-257
GD double maxc(D3 a) { return fmax(fmax(a.x, a.y), a.z); }
GD double luminance(D3 s) { return s.x * 0.212671 + s.y * 0.715160 + s.z * 0.072169; } // src/spectrum.h:33-35
GD double clamp01(double v) { return fmin(fmax(v, 0.0), 1.0); }
GD double sqr(double v) { return v * v; }
GD double pow5(double v) { double v2 = v * v; return v2 * v2 * v; }

// src/lajolla.h:42-55
GD double modulo_d(double a, double b) { double r = fmod(a, b); return (r < 0.0) ? r + b : r; }
GD int modulo_i(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }

struct Frame { D3 x, y, n; };
GD void coordinate_system(D3 n, D3 &a, D3 &b) { // src/frame.h:11-22
    if (n.z < (-1 + 1e-6)) { a = mk(0, -1, 0); b = mk(-1, 0, 0); }
    else {
        double aa = 1 / (1 + n.z);
        double bb = -n.x * n.y * aa;
        a = mk(1 - n.x * n.x * aa, bb, -n.x);
        b = mk(bb, 1 - n.y * n.y * aa, -n.y);
    }
}
GD Frame make_frame(D3 n) { Frame f; f.n = n; coordinate_system(n, f.x, f.y); return f; }
GD Frame neg(Frame f) { Frame r; r.x = -f.x; r.y = -f.y; r.n = -f.n; return r; }
GD D3 to_local(const Frame &f, D3 v) { return mk(dot(v, f.x), dot(v, f.y), dot(v, f.n)); }
GD D3 to_world(const Frame &f, D3 v) { return f.x * v.x + f.y * v.y + f.n * v.z; }

// PCG32 XSH-RR, src/pcg.h:18-66
struct Pcg { uint64_t state, inc; };
GD uint32_t pcg_next(Pcg &r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ULL + (r.inc | 1);
    uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = uint32_t(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
GD Pcg pcg_init(uint64_t stream) {
    Pcg s; s.state = 0U; s.inc = (stream << 1u) | 1u;
    pcg_next(s); s.state += 0x31e241f862a1fb5eULL; pcg_next(s);
    return s;
}
GD double pcg_real(Pcg &r) { // next_pcg32_real<double>: 32 random mantissa bits
    uint64_t u = ((uint64_t)pcg_next(r) << 20) | 0x3ff0000000000000ULL;
    return __longlong_as_double((long long)u) - 1.0;
}

} // namespace gd
