// sbvh_build.hpp -- host builder of the PRODUCTION walk's binary tree: SAH over triangle REFERENCES with spatial
// splits (Stich, Friedrich, Dietrich 2009, "Spatial Splits in Bounding Volume Hierarchies").  A reference is a triangle
// together with the box of the part of it that lies in the node; splitting a reference at a plane makes two whose boxes
// do not overlap across that plane, which is what a soup of triangles as large as their spacing (BASELINE C2 / C3) needs:
// with whole triangles the children of every low node overlap by a triangle's extent.
//
// The tree is not the canonical one of DESIGN.md 3.3 (that stays the oracle's, for the exact counters): by the tie rule
// of 3.4 a hit does not depend on the tree, a triangle reached through two references yields the same (t, id) twice.
// What must hold is ENCLOSURE -- every reference box contains the part of the triangle inside its node's region -- which
// the clipping below guarantees by computing intersections in double and rounding outwards.  The reference has no
// accelerator at all (core/api.rs:237: the name "bvh" is stored).
#pragma once
#include <cstdint>
#include <vector>

#include "bvh_build.hpp"

namespace pbrt_hip {

// A binary tree whose leaves hold runs of REFERENCES: nodes as BvhNode (leaf: offset = first reference, count in the low
// 16 bits of count_axis), reference r = triangle ref_tri[r] with box ref_lo / ref_hi (3 floats each).
struct RefBvh {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> ref_tri;
  std::vector<float> ref_lo, ref_hi;
  uint32_t depth = 0;
};

struct SbvhParams {
  float alpha = 1e-5f;        // try a spatial split when area(overlap of the best object split) / area(root) exceeds this
  float budget = 0.5f;        // extra references allowed, as a fraction of the triangle count
  int object_bins = 32;       // binned object SAH above sweep_below references, full sweep at and below
  int spatial_bins = 32;
  uint32_t sweep_below = 64;
  int widest_axis_only = 0;   // object splits on the axis of the widest centroid extent only (the canonical builder's rule)
  int low_side_first = 1;     // child 0 = the part on the low side of the split plane
  float spatial_bias = 1.f;   // a spatial split is taken when its SAH cost x this is below the best object split's
  float pad = 1e-5f;          // clipped boxes are widened by this fraction of the triangle's own extent (never beyond it)
};

// P: 3 * n_verts floats, idx: 3 * n_tris vertex indices.  Every leaf of the result holds ONE reference.
void build_sbvh(const float *P, const uint32_t *idx, uint32_t n_tris, const SbvhParams &prm, RefBvh *out);

// Global optimisation of a built tree whose leaves hold one reference each (PBRT_HIP_SCENE_OPTIMIZED_TREE): `passes` times, the
// `frac` largest interior nodes (by surface area; 1 = all) are taken out -- the sibling moves up -- and their two subtrees are put back
// where they add the least surface area along the path from the root, found by branch and bound (re-insertion, after Bittner, Hapala,
// Havran 2013, "Fast insertion-based optimization of bounding volume hierarchies").  Child 0 of every node is then the child whose
// centre is lower along the axis that separates the two centres most (the walk enters the nearer hit child and stacks the others in
// slot order).  Measured on BASELINE's meshes: 4-8 % fewer node steps and triangle tests per ray (tools/experiments/README.md).
void reinsert_optimize(RefBvh *t, int passes, float frac);

// the canonical tree seen as a reference tree (reference r = leaf slot r, boxes = the triangles' own bounds)
void refs_of_bvh(const Bvh &b, const float *P, const uint32_t *idx, RefBvh *out);

}  // namespace pbrt_hip
