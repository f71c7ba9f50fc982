// bvh_build.hpp -- host BVH builder of the render path: binned-SAH (16 buckets, <= 4 triangles per
// leaf) over a triangle soup, emitted directly in depth-first order as 32-byte nodes.
// The reference has no accelerator (only the name "bvh" is stored, core/api.rs:237,799-803); the
// node layout is the one SURVEY.md A3 fixes, the split rules are DESIGN.md section 3.3.
#pragma once
#include <cstdint>
#include <vector>

namespace pbrt_hip {

// 32 bytes; `offset` = first leaf slot (leaf) or index of the second child (interior; the first
// child is the node that follows).  word 7 = n_prims | axis << 16.
struct BvhNode {
  float lo[3];
  float hi[3];
  uint32_t offset;
  uint32_t count_axis;
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

struct Bvh {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> order;  // leaf slot -> triangle id
  uint32_t depth = 0;           // number of levels
};

// P: 3*n_verts floats, idx: 3*n_tris vertex indices
void build_bvh(const float *P, const uint32_t *idx, uint32_t n_tris, Bvh *out);

}  // namespace pbrt_hip
