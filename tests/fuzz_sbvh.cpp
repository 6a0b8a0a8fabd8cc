// fuzz_sbvh.cpp — manual robustness harness (not collected by pytest) for the tree build with spatial splits (host/sbvh.cpp) and its
// self-check: random triangle soups with long triangles, NaN / Inf vertices, coordinates of 1e30 and 1e-35, planar and identical
// triangles, budgets 0 .. 4, through gdpt_sbvh_check (nesting, reference budget, depth / stack bounds, coverage sampling).
// Build and run under the sanitizers (CPU only; ~10 min):
//   cd gradient-based-path-tracing_amd/csrc && for f in host/*.cpp capi_host.cpp; do g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off \
//       -fsanitize=address,undefined -fno-sanitize-recover=undefined -I. -I../../include -c $f -o /tmp/$(basename $f .cpp).o; done
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I../../include ../../tests/fuzz_sbvh.cpp /tmp/*.o -lz -lpthread -o /tmp/fuzz_sbvh && /tmp/fuzz_sbvh
// Round 3: 196 builds, no sanitizer report, no unexpected error (soups with NaN / Inf vertices are built, not sampled).
#include "gdpt.h"
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-10.f, 10.f);
    std::normal_distribution<float> Nrm(0.f, 0.4f);
    int32_t st[8];
    int fails = 0, runs = 0;
    for (int n : {1, 2, 3, 5, 17, 300, 2000}) {
        for (int variant = 0; variant < 7; variant++) {
            std::vector<float> t(9 * (size_t)n);
            for (int i = 0; i < n; i++) {
                float c[3] = {U(rng), U(rng), U(rng)};
                for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) t[9 * i + 3 * v + k] = (i < n / 50 + 1) ? U(rng) : c[k] + Nrm(rng);
            }
            if (variant == 1) for (size_t i = 0; i < t.size(); i += 97) t[i] = NAN;
            if (variant == 2) for (size_t i = 0; i < t.size(); i += 131) t[i] = INFINITY;
            if (variant == 3) for (auto &x : t) x *= 1e30f;
            if (variant == 4) for (auto &x : t) x *= 1e-35f;
            if (variant == 5) for (size_t i = 2; i < t.size(); i += 3) t[i] = 0.f;
            if (variant == 6) for (size_t i = 9; i < t.size(); i++) t[i] = t[i % 9];
            for (double budget : {0.0, 0.3, 1.0, 4.0}) {
                int rc = gdpt_sbvh_check(t.data(), n, budget, (variant == 1 || variant == 2) ? 0 : 8, st);
                runs++;
                if (rc != 0 && variant != 1 && variant != 2) { fails++; std::printf("n=%d variant=%d budget=%g: %s\n", n, variant, budget, gdpt_last_error()); }
            }
        }
    }
    std::printf("runs %d, unexpected errors %d\n", runs, fails);
    return fails != 0;
}
