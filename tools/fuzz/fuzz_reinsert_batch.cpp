// ASan / UBSan harness for the parallel re-insertion pass (pbrt_amd/csrc/reinsert_core.hpp through its host run,
// reinsert_batch.cpp): random, huge, tiny, identical, partly coincident and denormal-progression meshes; after EVERY pass the link
// tree must be a tree (every node reachable exactly once from the root, parent links pointing back, every child's box inside its
// parent's -- LinkTree::valid), the leaves a permutation of the references, and the pass must be deterministic (two runs: equal
// links).  Also reports the most nodes one search looked at (the visit cap must hold) and passes whose cost rose.
// usage: fuzz_reinsert_batch <seed> <iterations> [mode 0..5]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "bvh_build.hpp"
#include "reinsert_batch.hpp"

int main(int argc, char **argv) {
  std::mt19937 rng((unsigned)std::atoi(argv[1]));
  const int iters = std::atoi(argv[2]);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  unsigned long long worst_visits = 0, rose = 0, passes = 0, moves = 0;
  for (int it = 0; it < iters; it++) {
    const uint32_t n = rng() % 700;
    const int mode = argc > 3 ? std::atoi(argv[3]) : (int)(rng() % 6);
    std::vector<float> P(9 * (size_t)n);
    std::vector<uint32_t> idx(3 * (size_t)n);
    const float size = mode == 0 ? 0.05f : 1.f;
    for (uint32_t t = 0; t < n; t++) {
      float c[3] = {U(rng), U(rng), U(rng)};
      for (int v = 0; v < 3; v++)
        for (int a = 0; a < 3; a++) P[9 * (size_t)t + 3 * v + a] = (c[a] + size * U(rng)) * (mode == 1 ? 1e30f : (mode == 2 ? 1e-30f : 1.f));
    }
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
    if (mode == 3) for (size_t i = 0; i < P.size(); i++) P[i] = 0.25f;                                   // all identical
    if (mode == 4) for (uint32_t t = 0; t < n; t++) if (rng() % 3) for (int k = 0; k < 9; k++) P[9 * (size_t)t + k] = P[k];  // two thirds coincide with triangle 0
    if (mode == 5) { float s = 1.f; for (uint32_t t = 0; t < n; t++) { for (int k = 0; k < 9; k++) P[9 * (size_t)t + k] *= s; s *= 0.8f; } }  // down to denormals
    pbrt_hip::Bvh b;
    pbrt_hip::build_bvh(P.data(), idx.data(), n, &b);
    pbrt_hip::RefBvh rb;
    pbrt_hip::single_ref_tree(b, P.data(), idx.data(), &rb);
    if (rb.ref_tri.size() != n) { std::printf("single_ref_tree: %zu references for %u triangles\n", rb.ref_tri.size(), n); return 1; }
    if (n < 4) continue;
    pbrt_hip::LinkTree lt, twin;
    pbrt_hip::link_tree_of(rb, &lt);
    std::string why;
    if (!lt.valid(&why)) { std::printf("start tree invalid (mode %d, n %u): %s\n", mode, n, why.c_str()); return 1; }
    twin = lt;
    pbrt_hip::ReinsertBatchParams prm;
    prm.passes = 1;
    prm.mu = 1 + rng() % 3;
    prm.search.max_visits = (rng() & 1) ? 512u : 16u + rng() % 64;
    double cost = lt.cost();
    for (int pass = 0; pass < 6; pass++) {
      pbrt_hip::ReinsertBatchStats st;
      pbrt_hip::reinsert_batch_links(&lt, prm, &st);
      pbrt_hip::reinsert_batch_links(&twin, prm, nullptr);
      if (!lt.valid(&why)) { std::printf("pass %d broke the tree (mode %d, n %u): %s\n", pass, mode, n, why.c_str()); return 1; }
      if (lt.par != twin.par || lt.kid != twin.kid || lt.bx != twin.bx) { std::printf("pass %d is not deterministic (mode %d, n %u)\n", pass, mode, n); return 1; }
      if (st.max_visits > worst_visits) worst_visits = st.max_visits;
      // (a search finishes the subtree level it is on after reaching the cap: a few visits beyond it per level of the tree)
      if (st.max_visits > prm.search.max_visits + 256) { std::printf("a search looked at %llu nodes, cap %u\n", (unsigned long long)st.max_visits, prm.search.max_visits); return 1; }
      const double c2 = lt.cost();
      if (c2 > cost * (1.0 + 1e-5)) rose++;
      cost = c2;
      passes++;
      moves += st.applied;
    }
    std::vector<char> seen(n, 0);
    pbrt_hip::RefBvh out;
    pbrt_hip::ref_bvh_of(lt, rb, &out);
    if (out.ref_tri.size() != n) { std::printf("flattened tree has %zu references\n", out.ref_tri.size()); return 1; }
    for (uint32_t t : out.ref_tri) { if (t >= n || seen[t]) { std::printf("references are not a permutation (mode %d)\n", mode); return 1; } seen[t] = 1; }
  }
  std::printf("ok: %llu passes, %llu moves, cost rose in %llu passes, most nodes looked at by one search %llu\n", passes, moves, rose, worst_visits);
}
