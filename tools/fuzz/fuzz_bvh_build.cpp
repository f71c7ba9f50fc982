#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>
#include "bvh_build.hpp"
int main(int argc, char **argv) {
  std::mt19937 rng((unsigned)std::atoi(argv[1]));
  int iters = std::atoi(argv[2]);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (int it = 0; it < iters; it++) {
    uint32_t n = rng() % 400;
    std::vector<float> P(9 * (size_t)n);
    std::vector<uint32_t> idx(3 * (size_t)n);
    int mode = argc > 3 ? std::atoi(argv[3]) : (int)(rng() % 6);
    for (size_t i = 0; i < P.size(); i++) P[i] = U(rng) * (mode == 1 ? 1e30f : (mode == 2 ? 1e-30f : 1.f));
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
    if (mode == 3) for (size_t i = 0; i < P.size(); i++) P[i] = 0.25f;                      // all identical
    if (mode == 4 && n) for (int k = 0; k < 5; k++) P[rng() % P.size()] = std::numeric_limits<float>::quiet_NaN();
    if (mode == 5 && n) for (int k = 0; k < 5; k++) P[rng() % P.size()] = (rng() & 1) ? INFINITY : -INFINITY;
    pbrt_hip::Bvh b;
    pbrt_hip::build_bvh(P.data(), idx.data(), n, &b);
    if (b.order.size() != n) { std::printf("order size %zu != %u (mode %d)\n", b.order.size(), n, mode); return 1; }
    std::vector<char> seen(n, 0);
    for (uint32_t t : b.order) { if (t >= n || seen[t]) { std::printf("bad order (mode %d)\n", mode); return 1; } seen[t] = 1; }
    if (b.depth > 64 && mode < 4) { std::printf("depth %u (mode %d, n %u)\n", b.depth, mode, n); }
  }
  std::printf("ok\n");
}
