// ref_bvh.hpp -- a binary tree over triangle REFERENCES, the form the production walk's 4-wide nodes are collapsed from
// (capi.cpp make_quad_nodes_as) and the host run of the re-insertion pass works on (reinsert_batch.cpp).
#pragma once
#include <cstdint>
#include <vector>

#include "bvh_build.hpp"

namespace pbrt_hip {

// nodes as BvhNode in depth-first order (leaf: offset = first reference, count in the low 16 bits of count_axis; interior:
// first child = the next node, offset = second child); reference r = triangle ref_tri[r] with box ref_lo / ref_hi (3 floats each)
struct RefBvh {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> ref_tri;
  std::vector<float> ref_lo, ref_hi;
  uint32_t depth = 0;
};

// the canonical tree seen as a reference tree: reference r = leaf slot r, boxes = the triangles' own bounds
void refs_of_bvh(const Bvh &b, const float *P, const uint32_t *idx, RefBvh *out);
// the same with every leaf of several triangles opened into a subtree of single-triangle leaves (halved by position: the
// builder left a leaf's triangles sorted along its last split axis) -- the form re-insertion works on
void single_ref_tree(const Bvh &b, const float *P, const uint32_t *idx, RefBvh *out);

}  // namespace pbrt_hip
