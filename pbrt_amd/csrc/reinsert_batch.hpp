// reinsert_batch.hpp -- host run of the device's parallel re-insertion pass (reinsert_core.hpp) and the link form of a binary
// tree it works on.  Test / simulator infrastructure of the device builder: see reinsert_batch.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "reinsert_core.hpp"
#include "ref_bvh.hpp"

namespace pbrt_hip {

// A binary tree as the device holds it while optimising: interior nodes [0, n_int), leaf of reference r = n_int + r, root 0.
struct LinkTree {
  uint32_t n_int = 0;
  std::vector<uint32_t> par, kid;
  std::vector<unsigned long long> bx;
  // every node reachable exactly once from the root, links consistent, every child's box inside its parent's
  bool valid(std::string *why = nullptr) const;
  double cost() const;  // summed half surface area of the interior nodes
};

struct ReinsertBatchParams {
  reins::StopRule stop;         // the device loop's rule (passes at most, least moves per pass, visit budget): reinsert_core.hpp
  int passes = reins::StopRule().max_passes;  // (= stop.max_passes unless a caller wants fewer)
  uint32_t mu = 1;              // pass k searches the nodes x with (x + k) % mu == 0
  reins::Search search;
};
struct ReinsertBatchStats {
  uint64_t passes = 0, visits = 0, found = 0, applied = 0, max_visits = 0;
  uint64_t undone = 0;          // 1: the last pass raised the summed area and was undone
  double cost_before = 0, cost_after = 0;  // summed half surface area of the interior nodes
};

// `rb`: every leaf holds one reference
void link_tree_of(const RefBvh &rb, LinkTree *out);
// depth-first flattening (first child = the next node); references are renumbered in leaf order, `refs` supplies their triangles / boxes
void ref_bvh_of(const LinkTree &lt, const RefBvh &refs, RefBvh *out);
void refit_links(LinkTree *lt);
void reinsert_batch_links(LinkTree *lt, const ReinsertBatchParams &prm, ReinsertBatchStats *stats = nullptr);
void reinsert_optimize_batch(RefBvh *t, const ReinsertBatchParams &prm, ReinsertBatchStats *stats = nullptr);

}  // namespace pbrt_hip
