// ASan / UBSan harness for the spatial-split builder (pbrt_amd/csrc/sbvh_build.cpp): random, huge, tiny, identical and
// geometric-progression meshes (modes 0..3, 6; non-finite vertices are refused before any builder runs, so modes 4 / 5
// of fuzz_bvh_build are not repeated).  Checks: every triangle has at least one reference, every leaf one reference,
// every reference box lies inside the triangle's own box; then the re-insertion pass (reinsert_optimize) on the same tree.   usage: fuzz_sbvh_build <seed> <iterations> [mode]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "sbvh_build.hpp"
int main(int argc, char **argv) {
  std::mt19937 rng((unsigned)std::atoi(argv[1]));
  const int iters = std::atoi(argv[2]);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (int it = 0; it < iters; it++) {
    const uint32_t n = 1 + rng() % 600;
    std::vector<float> P(9 * (size_t)n);
    std::vector<uint32_t> idx(3 * (size_t)n);
    static const int modes[] = {0, 1, 2, 3, 6};
    const int mode = argc > 3 ? std::atoi(argv[3]) : modes[rng() % 5];
    for (size_t i = 0; i < P.size(); i++) P[i] = U(rng) * (mode == 1 ? 1e30f : (mode == 2 ? 1e-30f : 1.f));
    if (mode == 0) for (size_t t = 0; t < n; t++) for (int k = 1; k < 3; k++) for (int a = 0; a < 3; a++) P[9 * t + 3 * k + a] = P[9 * t + a] + 0.1f * U(rng);
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
    if (mode == 3) for (size_t i = 0; i < P.size(); i++) P[i] = 0.25f;  // all identical
    if (mode == 6) for (size_t t = 0; t < n; t++) { const float x = std::pow(0.5f, 0.5f * (float)t); for (int k = 0; k < 3; k++) { P[9 * t + 3 * k] = x * (1.f + 0.2f * k); P[9 * t + 3 * k + 1] = k == 2 ? 0.2f * x : 0.f; P[9 * t + 3 * k + 2] = 0.001f * (float)t; } }
    pbrt_hip::SbvhParams prm;
    prm.alpha = (rng() & 1) ? 0.f : 1e-5f;
    prm.budget = (float)(rng() % 4) * 0.5f;
    prm.spatial_bias = (rng() & 1) ? 1.f : 0.7f;
    pbrt_hip::RefBvh b;
    pbrt_hip::build_sbvh(P.data(), idx.data(), n, prm, &b);
    std::vector<char> seen(n, 0);
    for (size_t r = 0; r < b.ref_tri.size(); r++) {
      const uint32_t t = b.ref_tri[r];
      if (t >= n) { std::printf("bad triangle id (mode %d)\n", mode); return 1; }
      seen[t] = 1;
      for (int a = 0; a < 3; a++) {
        float lo = P[9 * t + a], hi = lo;
        for (int k = 1; k < 3; k++) { lo = std::fmin(lo, P[9 * t + 3 * k + a]); hi = std::fmax(hi, P[9 * t + 3 * k + a]); }
        if (b.ref_lo[3 * r + a] < lo || b.ref_hi[3 * r + a] > hi) { std::printf("reference box leaves its triangle's box (mode %d)\n", mode); return 1; }
      }
    }
    for (uint32_t t = 0; t < n; t++) if (!seen[t]) { std::printf("triangle %u has no reference (mode %d, n %u)\n", t, mode, n); return 1; }
    for (const auto &nd : b.nodes) if ((nd.count_axis & 0xffffu) > 1u) { std::printf("leaf with %u references\n", nd.count_axis & 0xffffu); return 1; }
    if (b.ref_tri.size() > (size_t)((double)n * (1.0 + prm.budget)) + 1) { std::printf("budget exceeded: %zu refs for %u triangles\n", b.ref_tri.size(), n); return 1; }
    // the re-insertion pass of PBRT_HIP_SCENE_OPTIMIZED_TREE on the same tree: the references survive as a permutation, every leaf holds
    // one, every node's box encloses its children's, child links stay inside the array, the depth field is the tree's depth
    {
      std::vector<uint32_t> before(b.ref_tri.begin(), b.ref_tri.end());
      pbrt_hip::reinsert_optimize(&b, 1 + (int)(rng() % 3), (rng() & 1) ? 1.0f : 0.3f);
      std::vector<uint32_t> after(b.ref_tri.begin(), b.ref_tri.end());
      std::sort(before.begin(), before.end());
      std::sort(after.begin(), after.end());
      if (before != after) { std::printf("re-insertion changed the references (mode %d, n %u)\n", mode, n); return 1; }
      if (b.ref_lo.size() != 3 * b.ref_tri.size() || b.ref_hi.size() != 3 * b.ref_tri.size()) { std::printf("reference boxes lost\n"); return 1; }
      struct It { uint32_t node, level; };
      std::vector<It> st = {{0u, 1u}};
      uint32_t depth = 0, leaves = 0;
      while (!st.empty() && !b.nodes.empty()) {
        const It it = st.back();
        st.pop_back();
        if (it.node >= b.nodes.size()) { std::printf("child link out of range\n"); return 1; }
        const auto &nd = b.nodes[it.node];
        if (it.level > depth) depth = it.level;
        if (nd.count_axis & 0xffffu) {
          if ((nd.count_axis & 0xffffu) != 1u || nd.offset >= b.ref_tri.size()) { std::printf("bad leaf after re-insertion\n"); return 1; }
          leaves++;
          for (int a = 0; a < 3; a++)
            if (nd.lo[a] > b.ref_lo[3 * (size_t)nd.offset + a] || nd.hi[a] < b.ref_hi[3 * (size_t)nd.offset + a]) { std::printf("leaf box does not hold its reference\n"); return 1; }
          continue;
        }
        const uint32_t kids[2] = {it.node + 1, nd.offset};
        for (uint32_t c : kids) {
          if (c >= b.nodes.size()) { std::printf("child link out of range\n"); return 1; }
          for (int a = 0; a < 3; a++)
            if (b.nodes[c].lo[a] < nd.lo[a] || b.nodes[c].hi[a] > nd.hi[a]) { std::printf("node box does not enclose its child (mode %d)\n", mode); return 1; }
          st.push_back({c, it.level + 1});
        }
      }
      if (!b.nodes.empty() && (leaves != b.ref_tri.size() || depth != b.depth)) { std::printf("leaves %u of %zu refs, depth %u vs %u\n", leaves, b.ref_tri.size(), depth, b.depth); return 1; }
    }
  }
  std::printf("ok\n");
}
