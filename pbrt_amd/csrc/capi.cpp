// capi.cpp -- the extern "C" boundary declared in include/pbrt_hip.h: scene flattening + upload,
// kernel launches, film assembly.  Replaces the (empty) body of PbrtAPI::world_end,
// /root/reference/src/core/api.rs:432-473.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <exception>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pbrt_hip.h"
#include "bvh_build.hpp"
#include "capi_internal.hpp"
#include "device_types.h"
#include "host_math.hpp"
#include "ref_bvh.hpp"
#include "reinsert_batch.hpp"
#include "scene_parser.hpp"

using namespace pbrt_hip;

namespace {

thread_local std::string g_err;

}  // namespace

namespace pbrt_hip {
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
const char *last_error_message() { return g_err.c_str(); }
}  // namespace pbrt_hip

namespace {

// number of 64x64 super-tiles a rank owns, and the grid of super-tiles
struct Shard {
  int32_t w, h;
  uint32_t stx, sty, total, n_local;
};
// the super-tiles of a rank over the pixel rectangle b (x0 y0 x1 y1): the cropped window, or the sample bounds of a wide filter
Shard make_shard_bounds(const int32_t b[4], uint32_t rank, uint32_t world) {
  Shard s;
  s.w = b[2] - b[0];
  s.h = b[3] - b[1];
  if (s.w < 0) s.w = 0;
  if (s.h < 0) s.h = 0;
  s.stx = (uint32_t)(s.w + 63) / 64;
  s.sty = (uint32_t)(s.h + 63) / 64;
  s.total = s.stx * s.sty;
  s.n_local = (world && rank < world && s.total > rank) ? (s.total - rank + world - 1) / world : 0;
  return s;
}
Shard make_shard(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world) {
  int32_t b[4];
  film_cropped_bounds(xres, yres, crop, b);
  return make_shard_bounds(b, rank, world);
}

// Film geometry of one render: the cropped window (film.rs:92-101), the sample bounds (film.rs:166-175) and whether the
// box filter has the default radius 0.5 -- then a sample lands in its own pixel and the two rectangles coincide -- or
// another one (samples reach pad = ceil(radius - 0.5) pixels beyond the window, fixed-point film).
struct FilmGeom {
  bool wide;
  float rx, ry;
  int32_t pad_x, pad_y;
  int32_t crop[4], sb[4];
  size_t crop_px() const { return (size_t)std::max(0, crop[2] - crop[0]) * (size_t)std::max(0, crop[3] - crop[1]); }
};
inline float filter_radius(float w) { return w == 0.f ? 0.5f : w; }
FilmGeom film_geom(const pbrt_hip_scene_desc &d, const pbrt_hip_render_desc &r) {
  FilmGeom g;
  g.rx = filter_radius(r.filter_xwidth);
  g.ry = filter_radius(r.filter_ywidth);
  g.wide = g.rx != 0.5f || g.ry != 0.5f;
  film_cropped_bounds(d.xres, d.yres, d.crop, g.crop);
  g.pad_x = g.pad_y = 0;
  for (int k = 0; k < 4; k++) g.sb[k] = g.crop[k];
  if (g.wide) {
    g.sb[0] = (int32_t)std::floor(((float)g.crop[0] + 0.5f) - g.rx);
    g.sb[1] = (int32_t)std::floor(((float)g.crop[1] + 0.5f) - g.ry);
    g.sb[2] = (int32_t)std::ceil(((float)g.crop[2] - 0.5f) + g.rx);
    g.sb[3] = (int32_t)std::ceil(((float)g.crop[3] - 0.5f) + g.ry);
    g.pad_x = std::max(0, (int32_t)std::ceil(g.rx - 0.5f));
    g.pad_y = std::max(0, (int32_t)std::ceil(g.ry - 0.5f));
    // an empty crop window has no sample bounds either: nothing is sampled for a film of no pixels (as the oracle: 0 rays)
    if (g.crop[2] <= g.crop[0] || g.crop[3] <= g.crop[1])
      for (int k = 0; k < 4; k++) g.sb[k] = g.crop[k];
  }
  return g;
}

// Scheduling thresholds of the traversal loop (kernels.hip trav_run).  They change only how lanes
// are interleaved, never a result; PBRT_HIP_MIN_WALKERS / PBRT_HIP_MIN_PARKED override them for
// tuning runs.
uint32_t tuning(const char *name, uint32_t dflt, long cap = 64) {
  const char *v = debug_knob(name);
  if (!v || !*v) return dflt;
  long x = std::strtol(v, nullptr, 10);
  return x < 0 ? 0u : (x > cap ? (uint32_t)cap : (uint32_t)x);
}
// min_walkers: 36 for deep trees (long walks: C3 +1 % over 32), 20 for shallow ones, where a frame is mostly shading and
// the shading stage should wait for more lanes (C4 +12 % over 36)
constexpr uint32_t kMinWalkers = 36, kMinWalkersShallow = 20, kShallowStackNeed = 16, kMinParked = 16;
// persistent one-wave workgroups of the render kernel per CU = what a CU holds at once: render_stack_plan (device_types.h:
// 20 with up to 32 LDS rows -- 5 waves per SIMD by the kernel's 96 VGPRs --, fewer with more rows); the kernel for scenes
// with spheres has the register budget of 3 waves per SIMD.  The grid is this x the device's CU count (hipDeviceProp_t)
constexpr uint32_t kRenderWavesPerCuSpheres = 12;
constexpr uint32_t kLeafRef = 0x80000000u;

// A vertex that a triangle uses and that is NaN or infinite would send the builders' bucket index out of range:
// such input is refused at the boundary.  Returns the first offending vertex, or -1.
long long first_non_finite_vertex(const float *P, const uint32_t *idx, uint32_t n_tris) {
  for (size_t i = 0; i < 3 * (size_t)n_tris; i++) {
    const float *v = P + 3 * (size_t)idx[i];
    if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2])) return (long long)idx[i];
  }
  return -1;
}

// "Children in parent" form of the binary tree for the kernels: one 64-byte record per INTERIOR
// node with the boxes and references of its two children.  ref = interior index
// (dense numbering of interior nodes in depth-first order) or kLeafRef | n_prims << 24 | first slot.
struct PairNodes {
  std::vector<uint4> q;  // 4 per interior node
  uint32_t root_ref = 0xffffffffu;
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
};
bool make_pair_nodes(const Bvh &b, PairNodes *out, std::string *why) {
  const size_t n = b.nodes.size();
  if (n == 0) return true;
  if (b.order.size() > (1u << 24)) { *why = "more than 2^24 triangles (leaf references hold a 24-bit slot)"; return false; }
  // Record numbering.  The memory system past L2 serves random requests in 128-byte lines at a rate
  // that does not depend on how many of the 128 bytes are used (tools/ubench/gather_wide.hip), so
  // the two 64-byte records of SIBLING interior nodes are placed in one line: fetching the near
  // child's record brings the far one along.  Sibling pairs start at even indices; groups follow
  // each other in depth-first order.  PBRT_HIP_NODE_LAYOUT=dfs restores plain depth-first numbering.
  std::vector<uint32_t> interior_index(n, 0);
  uint32_t n_int = 0;
  const char *layout = debug_knob("PBRT_HIP_NODE_LAYOUT");
  if (layout && std::string(layout) == "dfs") {
    for (size_t i = 0; i < n; i++)
      if ((b.nodes[i].count_axis & 0xffffu) == 0) interior_index[i] = n_int++;
  } else if ((b.nodes[0].count_axis & 0xffffu) == 0) {
    std::vector<uint32_t> todo = {0};
    interior_index[0] = 0;
    n_int = 2;  // the root has its line to itself
    while (!todo.empty()) {
      const uint32_t p = todo.back();
      todo.pop_back();
      const uint32_t c0 = p + 1, c1 = b.nodes[p].offset;
      const bool i0 = (b.nodes[c0].count_axis & 0xffffu) == 0, i1 = (b.nodes[c1].count_axis & 0xffffu) == 0;
      if (i0 && i1) {
        n_int = (n_int + 1u) & ~1u;
        interior_index[c0] = n_int;
        interior_index[c1] = n_int + 1;
        n_int += 2;
      } else if (i0) {
        interior_index[c0] = n_int++;
      } else if (i1) {
        interior_index[c1] = n_int++;
      }
      if (i1) todo.push_back(c1);
      if (i0) todo.push_back(c0);
    }
  }
  auto ref_of = [&](uint32_t i) -> uint32_t {
    const BvhNode &c = b.nodes[i];
    const uint32_t cnt = c.count_axis & 0xffffu;
    return cnt ? (kLeafRef | (cnt << 24) | c.offset) : interior_index[i];
  };
  auto as_u = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  out->q.assign(4 * (size_t)n_int, make_uint4(0u, 0u, 0u, 0u));
  for (size_t i = 0; i < n; i++) {
    const BvhNode &p = b.nodes[i];
    if (p.count_axis & 0xffffu) continue;
    const BvhNode &c0 = b.nodes[i + 1], &c1 = b.nodes[p.offset];
    uint4 *q = &out->q[4 * (size_t)interior_index[i]];
    q[0] = make_uint4(as_u(c0.lo[0]), as_u(c0.lo[1]), as_u(c0.lo[2]), as_u(c0.hi[0]));
    q[1] = make_uint4(as_u(c0.hi[1]), as_u(c0.hi[2]), as_u(c1.lo[0]), as_u(c1.lo[1]));
    q[2] = make_uint4(as_u(c1.lo[2]), as_u(c1.hi[0]), as_u(c1.hi[1]), as_u(c1.hi[2]));
    q[3] = make_uint4(ref_of((uint32_t)i + 1), ref_of(p.offset), p.count_axis >> 16, 0u);
  }
  out->root_ref = ref_of(0);
  for (int a = 0; a < 3; a++) { out->root_lo[a] = b.nodes[0].lo[a]; out->root_hi[a] = b.nodes[0].hi[a]; }
  return true;
}

// 4-wide, QUANTISED form of the same tree for the production walk.  Every quad node is a binary
// interior node collapsed with its interior children (2..4 children); the children's boxes are
// stored as 8-bit coordinates on the node's own grid (origin = the node's lower corner, one
// power-of-two cell size per axis), rounded outwards, so a node with four children is 64 bytes:
//   {origin.x origin.y origin.z  cell.x}                    cell sizes as f32 (powers of two)
//   {qlo.x[4]  qlo.y[4]  qlo.z[4]  qhi.x[4]}               one byte per child
//   {qhi.y[4]  qhi.z[4]  cell.y  cell.z}
//   {ref[4]}                                                interior child: its byte offset in this array (node x 64);
//                                                           leaf child: 1<<31 | count<<24 | first leaf slot
// plane = origin + q * cell (a real number): the builder checks in exact (double) arithmetic that
// every decoded box contains the true one, so the walk visits a superset of the exact walk's nodes and the
// RESULT is unchanged (tie rule of DESIGN.md 3.4).  Why: the loop is bound by the bytes it moves
// from L2 to L1, and this form moves ~2.9 KB per ray instead of ~4.9 KB.
// Unused child slots: qlo = 255, qhi = 0 (inverted), ref kEmptyLeafRef (a leaf without triangles: device_types.h).
struct QuadNodes {
  std::vector<uint4> q;     // 4 per node
  uint32_t stack_need = 0;  // most entries the walk can hold: max over root-to-leaf paths of sum(children - 1)
  std::vector<float> exact;  // diagnostics (tools/walk_sim.py): the children's boxes before quantisation, 24 floats per node
};
// A child of a quad node while it is being assembled: a node of the binary tree, or (split_leaves) one
// triangle of a leaf that was expanded into a quad node of single-triangle children.
struct QuadChild {
  float lo[3], hi[3];
  uint32_t ref;        // final ref (leaf) or 0 with `node` / `leaf_node` set
  uint32_t node;       // binary interior node to recurse into, or 0xffffffff
  uint32_t leaf_node;  // binary leaf to expand into its own quad node, or 0xffffffff
};
enum Collapse { kCollapsePlain = 0, kCollapseGreedy = 1, kCollapseDp = 2 };
// `b`: the binary tree over triangle references (the canonical tree through refs_of_bvh, or the optimised single-triangle tree of
// single_ref_tree + reinsert_optimize_batch); slot_of_ref[r] = slot of reference r's triangle in the leaf-ordered triangle records (null: r itself).
void make_quad_nodes_as(const RefBvh &b, const uint32_t *slot_of_ref, bool split_leaves, Collapse how, QuadNodes *out) {
  const bool greedy = how != kCollapsePlain;
  if (b.nodes.empty() || (b.nodes[0].count_axis & 0xffffu) != 0) return;  // no tree, or the root is a leaf
  auto as_u = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  // Which descendants become the (up to four) children of a quad node?  Greedy: open the child with the largest
  // surface area while the result fits four slots.  Dp instead minimises, by dynamic
  // programming over the binary tree (after Ylitie, Karras, Laine 2017, section 3.2), the expected work of a walk:
  // every child of a quad node R is reached with the probability of its box AS R's 8-BIT GRID HOLDS IT (about one
  // cell of R wider per axis -- a small child of a large node gets a coarse box), a reached interior child costs one
  // node step, a reached triangle c_tri.  R is the ancestor at binary distance d = 1..3 of the node in question:
  //   F(n, k, d) = least expected work inside subtree n when n may occupy up to k child slots of R
  //              = min( present n as ONE child: Aq(n, R) * (c_tri * #triangles)   for a plain leaf,
  //                                             Aq(n, R) + G(n)                   else (one step at n, plus below),
  //                     open n (k >= 2, d < 3): min_{k1+k2=k} F(l, k1, d+1) + F(r, k2, d+1) )
  // with G(n) = min_{k1+k2=4} F(l,k1,1) + F(r,k2,1) the work below n as a quad node of its own (areas are
  // unconditional reach probabilities up to the common factor 1/A(root), as in the SAH).
  // Measured (c_tri = 2): C3 40.2 instead of 41.0 fetches per ray but a stack bound of 41 (overflow variant): -1 %;
  // C2 +2 %.  Without the quantisation term the same programme made 11.6 % fewer nodes and C3 6 % slower.
  const bool use_dp = how == kCollapseDp;
  static const float c_tri = debug_knob("PBRT_HIP_COLLAPSE_CTRI") ? (float)std::atof(debug_knob("PBRT_HIP_COLLAPSE_CTRI")) : 2.0f;
  const size_t nn = b.nodes.size();
  std::vector<float> F;            // F[(4 * n + (k - 1)) * 3 + (d - 1)]
  std::vector<float> G;            // work below n as a quad node of its own (interior nodes and splittable leaves)
  std::vector<uint32_t> parent;
  auto Fi = [](size_t n, uint32_t k, uint32_t d) { return (4 * n + (k - 1)) * 3 + (d - 1); };
  auto anc = [&](uint32_t n, uint32_t d) { while (d-- && parent[n] != 0xffffffffu) n = parent[n]; return n; };
  // surface area of box (lo, hi) as the grid of quad node q holds it: about one cell wider per axis
  auto area_q = [&](const float *lo, const float *hi, const BvhNode &q) {
    float dd[3];
    for (int a = 0; a < 3; a++) {
      const float ext = q.hi[a] - q.lo[a];
      int e = -126;
      if (ext > 0.f) { std::frexp(ext / 255.0f, &e); if (e < -126) e = -126; }
      dd[a] = (hi[a] - lo[a]) + std::ldexp(1.0f, e);
    }
    return (dd[0] * dd[1] + dd[0] * dd[2]) + dd[1] * dd[2];
  };
  auto slot_of = [&](uint32_t r) { return slot_of_ref ? slot_of_ref[r] : r; };
  auto tri_box = [&](uint32_t r, float lo[3], float hi[3]) {  // box of reference r
    for (int a = 0; a < 3; a++) { lo[a] = b.ref_lo[3 * (size_t)r + a]; hi[a] = b.ref_hi[3 * (size_t)r + a]; }
  };
  auto one_cost = [&](uint32_t n, uint32_t d) {  // n presented as ONE child of its ancestor at distance d
    const BvhNode &nd = b.nodes[n];
    const uint32_t cnt = nd.count_axis & 0xffffu;
    const float reach = area_q(nd.lo, nd.hi, b.nodes[anc(n, d)]);
    if (cnt && !(split_leaves && cnt >= 2 && cnt <= 4)) return reach * c_tri * (float)cnt;  // plain leaf: its triangles are tested
    return reach + G[n];  // a quad node of its own: one step when reached, plus what lies below
  };
  if (use_dp) {
    F.assign(12 * nn, 0.f);
    G.assign(nn, 0.f);
    parent.assign(nn, 0xffffffffu);
    for (size_t i = 0; i < nn; i++)
      if ((b.nodes[i].count_axis & 0xffffu) == 0) { parent[i + 1] = (uint32_t)i; parent[b.nodes[i].offset] = (uint32_t)i; }
    for (size_t i = nn; i-- > 0;) {  // children have larger indices than their parent (depth-first order)
      const BvhNode &n = b.nodes[i];
      const uint32_t cnt = n.count_axis & 0xffffu;
      if (cnt) {
        const bool splittable = split_leaves && cnt >= 2 && cnt <= 4;
        if (splittable) {  // as a quad node of its own its triangles sit on ITS grid
          float g = 0.f;
          for (uint32_t j = 0; j < cnt; j++) { float lo[3], hi[3]; tri_box(n.offset + j, lo, hi); g += c_tri * area_q(lo, hi, n); }
          G[i] = g;
        }
        for (uint32_t d = 1; d <= 3; d++) {
          const float one = one_cost((uint32_t)i, d);
          float opened = std::numeric_limits<float>::infinity();
          if (splittable) {  // opened: its triangles are direct children of the ancestor
            opened = 0.f;
            const BvhNode &q = b.nodes[anc((uint32_t)i, d)];
            for (uint32_t j = 0; j < cnt; j++) { float lo[3], hi[3]; tri_box(n.offset + j, lo, hi); opened += c_tri * area_q(lo, hi, q); }
          }
          for (uint32_t k = 1; k <= 4; k++) F[Fi(i, k, d)] = (splittable && k >= cnt) ? std::min(one, opened) : one;
        }
      } else {
        const size_t l = i + 1, r = n.offset;
        auto dist = [&](uint32_t k, uint32_t d) {
          float best = std::numeric_limits<float>::infinity();
          for (uint32_t k1 = 1; k1 < k; k1++) best = std::min(best, F[Fi(l, k1, d)] + F[Fi(r, k - k1, d)]);
          return best;
        };
        G[i] = dist(4, 1);
        for (uint32_t d = 1; d <= 3; d++) {
          const float one = one_cost((uint32_t)i, d);
          F[Fi(i, 1, d)] = one;
          for (uint32_t k = 2; k <= 4; k++) F[Fi(i, k, d)] = d < 3 ? std::min(one, dist(k, d + 1)) : one;
        }
      }
    }
  }
  if (use_dp && std::getenv("PBRT_HIP_REINSERT_VERBOSE")) {
    const BvhNode &r = b.nodes[0];
    const float dx = r.hi[0] - r.lo[0], dy = r.hi[1] - r.lo[1], dz = r.hi[2] - r.lo[2];
    std::fprintf(stderr, "collapse: expected work below the root G = %.4f root areas\n", G[0] / ((dx * dy + dx * dz) + dy * dz));
  }
  struct Item { uint32_t node, quad, path; bool is_leaf; };  // path = stack entries held above this node
  std::vector<Item> todo = {{0u, 0u, 0u, false}};
  out->q.assign(4, make_uint4(0, 0, 0, 0));
  while (!todo.empty()) {
    const Item it = todo.back();
    todo.pop_back();
    const BvhNode &me = b.nodes[it.node];
    QuadChild kids[4];
    int nk = 0;
    auto add_node = [&](uint32_t c) {
      const BvhNode &n = b.nodes[c];
      QuadChild k;
      for (int a = 0; a < 3; a++) { k.lo[a] = n.lo[a]; k.hi[a] = n.hi[a]; }
      const uint32_t cnt = n.count_axis & 0xffffu;
      k.ref = cnt ? (kLeafRef | (cnt << 24) | slot_of(n.offset)) : 0u;  // (a run of several references: consecutive slots)
      k.node = cnt ? 0xffffffffu : c;
      // a leaf of 2..4 triangles becomes a quad node of single triangles: their boxes are then tested in
      // the node step and each leaf pass tests exactly one triangle per parked lane
      k.leaf_node = (split_leaves && cnt >= 2 && cnt <= 4) ? c : 0xffffffffu;
      kids[nk++] = k;
    };
    auto tri_child = [&](uint32_t r) {
      QuadChild k;
      tri_box(r, k.lo, k.hi);
      k.ref = kLeafRef | (1u << 24) | slot_of(r);
      k.node = k.leaf_node = 0xffffffffu;
      return k;
    };
    if (it.is_leaf) {  // expand a leaf: one child per triangle, boxed by its own bounds
      const uint32_t cnt = me.count_axis & 0xffffu;
      for (uint32_t j = 0; j < cnt; j++) kids[nk++] = tri_child(me.offset + j);
    } else {
      // Greedy collapse: start from the two children of the binary node and keep opening the child with the
      // largest surface area (an interior node into its two children, a small leaf into its triangles) while
      // the result still fits four slots.
      if (use_dp) {
        // follow the minimising choices: node c with k slots at distance d is opened or presented as one child
        struct Open { uint32_t c, k, d; };
        std::vector<Open> st;
        auto split = [&](uint32_t c, uint32_t k, uint32_t d) {  // children of c share k slots at distance d; ties: the most even split
          const size_t l = c + 1, r = b.nodes[c].offset;
          uint32_t bk = 1;
          float best = std::numeric_limits<float>::infinity();
          for (uint32_t k1 = 1; k1 < k; k1++) {
            const float v = F[Fi(l, k1, d)] + F[Fi(r, k - k1, d)];
            if (v < best || (v == best && std::abs((int)(2 * k1) - (int)k) < std::abs((int)(2 * bk) - (int)k))) { best = v; bk = k1; }
          }
          st.push_back({(uint32_t)r, k - bk, d});
          st.push_back({(uint32_t)l, bk, d});
        };
        split(it.node, 4u, 1u);
        while (!st.empty()) {
          const Open o = st.back();
          st.pop_back();
          const BvhNode &n = b.nodes[o.c];
          const uint32_t cnt = n.count_axis & 0xffffu;
          const float one = one_cost(o.c, o.d);
          if (cnt) {
            if (split_leaves && cnt >= 2 && cnt <= 4 && o.k >= cnt && F[Fi(o.c, o.k, o.d)] < one) {
              for (uint32_t j = 0; j < cnt; j++) kids[nk++] = tri_child(n.offset + j);  // opened: its triangles are direct children
            } else {
              add_node(o.c);
            }
          } else if (o.k >= 2u && o.d < 3u && F[Fi(o.c, o.k, o.d)] < one) {
            split(o.c, o.k, o.d + 1u);
          } else {
            add_node(o.c);
          }
        }
      } else {
      add_node(it.node + 1);
      add_node(me.offset);
      auto area = [](const QuadChild &k) {
        const float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
        return (dx * dy + dx * dz) + dy * dz;
      };
      for (;;) {
        int best = -1;
        float best_area = -1.f;
        for (int k = 0; k < nk; k++) {
          const QuadChild &c = kids[k];
          const uint32_t grow = c.node != 0xffffffffu ? 1u : (c.leaf_node != 0xffffffffu ? (b.nodes[c.leaf_node].count_axis & 0xffffu) - 1u : 99u);
          if (greedy && (uint32_t)nk + grow <= 4u && area(c) > best_area) { best = k; best_area = area(c); }
        }
        if (best < 0) break;
        const QuadChild c = kids[best];
        kids[best] = kids[--nk];
        if (c.node != 0xffffffffu) {
          add_node(c.node + 1);
          add_node(b.nodes[c.node].offset);
        } else {
          const BvhNode &lf = b.nodes[c.leaf_node];
          for (uint32_t j = 0; j < (lf.count_axis & 0xffffu); j++) kids[nk++] = tri_child(lf.offset + j);
        }
      }
      if (!greedy) {  // the plain collapse: both children opened once
        nk = 0;
        const uint32_t two[2] = {it.node + 1, me.offset};
        for (uint32_t c : two) {
          if ((b.nodes[c].count_axis & 0xffffu) == 0) { add_node(c + 1); add_node(b.nodes[c].offset); }
          else add_node(c);
        }
      }
      }
    }
    const uint32_t path = it.path + (uint32_t)(nk - 1);
    if (path > out->stack_need) out->stack_need = path;
    uint32_t ebyte[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++) {
      const float origin = me.lo[a], extent = me.hi[a] - me.lo[a];
      // smallest power-of-two cell with 255 cells covering the extent (bumped while rounding pushes a plane past 255)
      int e = -126;
      if (extent > 0.f) {
        std::frexp(extent / 255.0f, &e);  // extent/255 = m * 2^e, m in [0.5, 1)  =>  2^e >= extent/255
        if (e < -126) e = -126;
      }
      for (;; e++) {
        const float cell = std::ldexp(1.0f, e);
        bool ok = true;
        uint32_t lo_bytes = 0, hi_bytes = 0;
        for (int k = 0; k < 4 && ok; k++) {
          if (k >= nk) { lo_bytes |= 255u << (8 * k); continue; }
          const QuadChild &c = kids[k];
          int ql = (int)std::floor((c.lo[a] - origin) / cell), qh = (int)std::ceil((c.hi[a] - origin) / cell);
          if (ql < 0) ql = 0;
          if (qh < 0) qh = 0;
          // enclosure checked in exact arithmetic: origin + q * cell fits a double without rounding
          const double o64 = origin, c64 = cell;
          while (ql > 0 && o64 + ql * c64 > (double)c.lo[a]) ql--;
          while (qh <= 255 && o64 + qh * c64 < (double)c.hi[a]) qh++;
          if (ql > 255 || qh > 255 || o64 + ql * c64 > (double)c.lo[a]) { ok = false; break; }
          lo_bytes |= (uint32_t)ql << (8 * k);
          hi_bytes |= (uint32_t)qh << (8 * k);
        }
        if (ok) { ebyte[a] = (uint32_t)(e + 127); qlo[a] = lo_bytes; qhi[a] = hi_bytes; break; }
      }
    }
    uint32_t ref[4];
    for (int k = 0; k < 4; k++) {
      if (k >= nk) { ref[k] = kEmptyLeafRef; continue; }
      const QuadChild &c = kids[k];
      if (c.node == 0xffffffffu && c.leaf_node == 0xffffffffu) {
        ref[k] = c.ref;
      } else {
        const uint32_t quad = (uint32_t)(out->q.size() / 4);
        ref[k] = quad * 64u;  // an interior child's ref is its byte offset in the node array: no shift in the walk
        out->q.resize(out->q.size() + 4, make_uint4(0, 0, 0, 0));
        // below child k the walk holds the entries of this path minus the ones already popped: bound by path
        if (c.node != 0xffffffffu) todo.push_back({c.node, quad, path, false});
        else todo.push_back({c.leaf_node, quad, path, true});
      }
    }
    out->exact.resize(out->q.size() / 4 * 24, 0.f);
    for (int k = 0; k < 4; k++)
      for (int a = 0; a < 3; a++) {
        out->exact[(size_t)it.quad * 24 + k * 6 + a] = k < nk ? kids[k].lo[a] : std::numeric_limits<float>::infinity();
        out->exact[(size_t)it.quad * 24 + k * 6 + 3 + a] = k < nk ? kids[k].hi[a] : -std::numeric_limits<float>::infinity();
      }
    uint4 *q = &out->q[4 * (size_t)it.quad];
    // the three cell sizes as f32 bit patterns (powers of two: exponent byte << 23), ready to be multiplied by 1 / d
    q[0] = make_uint4(as_u(me.lo[0]), as_u(me.lo[1]), as_u(me.lo[2]), ebyte[0] << 23);
    q[1] = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
    q[2] = make_uint4(qhi[1], qhi[2], ebyte[1] << 23, ebyte[2] << 23);
    q[3] = make_uint4(ref[0], ref[1], ref[2], ref[3]);
  }
}

// A-B aid (PBRT_HIP_QUAD_LAYOUT=pre|pre_big behind the debug switch): renumber the quad nodes in depth-first PRE-order, so
// that a node and the first interior child visited after it share a 128-byte line (the collapse above keeps SIBLINGS
// together instead: a family is one or two lines).  pre: children in slot order; pre_big: the child with the largest box
// first.  The tree, and so every result, is unchanged.
void relayout_quads(QuadNodes *q, bool big_first) {
  const size_t n = q->q.size() / 4;
  if (n < 2) return;
  auto area = [&](size_t node, int k) {
    const uint4 *w = &q->q[4 * node];
    auto byte = [](uint32_t v, int i) { return (float)((v >> (8 * i)) & 0xffu); };
    float cx, cy, cz;
    std::memcpy(&cx, &w[0].w, 4); std::memcpy(&cy, &w[2].z, 4); std::memcpy(&cz, &w[2].w, 4);
    const float dx = (byte(w[1].w, k) - byte(w[1].x, k)) * cx, dy = (byte(w[2].x, k) - byte(w[1].y, k)) * cy, dz = (byte(w[2].y, k) - byte(w[1].z, k)) * cz;
    return (dx * dy + dx * dz) + dy * dz;
  };
  std::vector<uint32_t> new_of(n, 0xffffffffu), order;
  order.reserve(n);
  std::vector<uint32_t> st = {0u};
  while (!st.empty()) {
    const uint32_t me = st.back();
    st.pop_back();
    new_of[me] = (uint32_t)order.size();
    order.push_back(me);
    const uint32_t ref[4] = {q->q[4 * (size_t)me + 3].x, q->q[4 * (size_t)me + 3].y, q->q[4 * (size_t)me + 3].z, q->q[4 * (size_t)me + 3].w};
    int ks[4], m = 0;
    for (int k = 0; k < 4; k++)
      if (!(ref[k] & kLeafRef)) ks[m++] = k;
    if (big_first) std::sort(ks, ks + m, [&](int a, int b) { return area(me, a) > area(me, b); });
    for (int i = m - 1; i >= 0; i--) st.push_back(ref[ks[i]] / 64u);  // (the first of ks is popped next: it follows its parent)
  }
  std::vector<uint4> nq(q->q.size());
  std::vector<float> ne(q->exact.size());
  for (size_t i = 0; i < n; i++) {
    const size_t o = order[i];
    for (int w = 0; w < 4; w++) nq[4 * i + w] = q->q[4 * o + w];
    uint32_t *r = &nq[4 * i + 3].x;
    for (int k = 0; k < 4; k++)
      if (!(r[k] & kLeafRef)) r[k] = new_of[r[k] / 64u] * 64u;
    if (!ne.empty()) std::memcpy(&ne[24 * i], &q->exact[24 * o], 24 * sizeof(float));
  }
  q->q.swap(nq);
  q->exact.swap(ne);
}

// The tree the walk gets: the quantisation-aware dynamic-programming collapse for trees of 1024 triangles and more, the
// greedy one below that.  Measured (r02f kernel): dp is 2.5 % faster on C3 (433 k instead of 488 k nodes, 40.2 instead of 41.0
// fetches per ray), 1.6 % on C2, 3.1 % on the 12 M-triangle workload, but 2 % slower on C4's 19-node tree; its build takes a
// third longer.  (Until r02e its deeper stack bound -- C3: 41 instead of 38 -- cost it the overflow variant of the walk, which
// every big tree takes now anyway.)  PBRT_HIP_COLLAPSE=dp|greedy|plain overrides (PBRT_HIP_GREEDY_COLLAPSE=0 = plain).
constexpr uint32_t kDpCollapseMinTris = 1024;
void make_quad_nodes(const RefBvh &b, const uint32_t *slot_of_ref, bool split_leaves, QuadNodes *out) {
  const char *c = debug_knob("PBRT_HIP_COLLAPSE");
  const char *g = debug_knob("PBRT_HIP_GREEDY_COLLAPSE");
  Collapse how = b.ref_tri.size() >= kDpCollapseMinTris ? kCollapseDp : kCollapseGreedy;
  if ((g && g[0] == '0') || (c && std::strcmp(c, "plain") == 0)) how = kCollapsePlain;
  else if (c && std::strcmp(c, "dp") == 0) how = kCollapseDp;
  else if (c && std::strcmp(c, "greedy") == 0) how = kCollapseGreedy;
  make_quad_nodes_as(b, slot_of_ref, split_leaves, how, out);
  if (const char *l = debug_knob("PBRT_HIP_QUAD_LAYOUT")) {
    if (std::strcmp(l, "pre") == 0) relayout_quads(out, false);
    else if (std::strcmp(l, "pre_big") == 0) relayout_quads(out, true);
  }
}

// The production walk's 4-wide tree of a triangle soup, built on the host.  `tree` picks the binary tree it is collapsed from:
// kTreeCanonical = the canonical binned-SAH tree `canon` (the oracle's tree); kTreeReinsert = that tree with its
// leaves opened into single triangles and optimised by the DEVICE builder's parallel re-insertion pass run on the host
// (reinsert_batch.cpp: the same functions, reinsert_core.hpp).  Either way a leaf child's slot refers to the triangle records in
// `canon`'s leaf order.  (Round 3's host-only builders -- spatial splits, sequential re-insertion -- are records now:
// tools/experiments/r03_host_tree_builders/.)
enum ProductionTree : uint32_t { kTreeCanonical = 0, kTreeReinsert = 2 };  // (= PBRT_HIP_TREE_*)
ReinsertBatchParams reinsert_batch_params() {
  ReinsertBatchParams p;  // (passes, stop rule and least tree size: reins::StopRule, shared with the device loop of bvh_gpu.hip)
  if (const char *v = debug_knob("PBRT_HIP_REINSERT")) p.passes = std::atoi(v);
  if (const char *v = debug_knob("PBRT_HIP_REINSERT_MIN_TRIS")) p.stop.min_tris = (uint32_t)std::max(8, std::atoi(v));
  if (const char *v = debug_knob("PBRT_HIP_REINSERT_MU")) p.mu = (uint32_t)std::max(1, std::atoi(v));
  if (const char *v = debug_knob("PBRT_HIP_REINSERT_VISITS")) p.search.max_visits = (uint32_t)std::max(1, std::atoi(v));
  if (const char *v = debug_knob("PBRT_HIP_REINSERT_MIN_REL")) p.search.min_rel = (float)std::atof(v);
  if (const char *v = debug_knob("PBRT_HIP_REINSERT_QK")) p.search.qk = (float)std::atof(v);
  if (const char *v = debug_knob("PBRT_HIP_REINSERT_QW")) p.search.qw = (float)std::atof(v);
  return p;
}
ProductionTree production_tree_default() {
  const char *v = debug_knob("PBRT_HIP_TREE");
  if (v && std::strcmp(v, "reinsert") == 0) return kTreeReinsert;
  return kTreeCanonical;
}
void build_production_quads(const Bvh &canon, const float *P, const uint32_t *idx, uint32_t n_tris, ProductionTree tree,
                            bool split_leaves, QuadNodes *out, uint32_t *n_refs = nullptr) {
  RefBvh rb;
  if (tree == kTreeReinsert && n_tris >= 2) {
    single_ref_tree(canon, P, idx, &rb);
    const ReinsertBatchParams rp = reinsert_batch_params();
    if (n_tris >= rp.stop.min_tris) reinsert_optimize_batch(&rb, rp);  // (smaller trees stay as built, as on the device)
    if (std::getenv("PBRT_HIP_REINSERT_VERBOSE")) {
      LinkTree lt;
      link_tree_of(rb, &lt);
      std::fprintf(stderr, "tree %u: summed interior half-area %.6g, depth %u\n", (unsigned)tree, lt.cost(), rb.depth);
    }
    std::vector<uint32_t> slot_of_tri(n_tris), slot_of_ref(rb.ref_tri.size());
    for (uint32_t s = 0; s < n_tris; s++) slot_of_tri[canon.order[s]] = s;
    for (size_t r = 0; r < rb.ref_tri.size(); r++) slot_of_ref[r] = slot_of_tri[rb.ref_tri[r]];
    make_quad_nodes(rb, slot_of_ref.data(), true, out);
  } else {
    refs_of_bvh(canon, P, idx, &rb);
    make_quad_nodes(rb, nullptr, split_leaves, out);
  }
  if (n_refs) *n_refs = (uint32_t)rb.ref_tri.size();
}

// The counter flags (PBRT_HIP_FLAG_COUNTERS, counters of pbrt_hip_intersect) count the CANONICAL walk: the oracle's binary
// tree of DESIGN.md 3.3.  A host-built scene has it; a device-built one gets it here on first use -- vertex / index buffers
// read back from the device, the host builder (one core, about a second for 1M triangles: a measurement aid, off the
// product's path), child-pair records and triangle records in the canonical leaf order uploaded beside the production arrays.
int ensure_canonical(pbrt_hip_scene *s) {
  if (s->canonical_ready) return PBRT_HIP_OK;
  if (!s->gpu_built) { s->dev_exact = s->dev; s->canonical_ready = true; return PBRT_HIP_OK; }
  const auto t0 = std::chrono::steady_clock::now();
  const uint32_t nt = s->n_prims;  // (triangles + the spheres' proxy triangles)
  std::vector<float> P(s->d_P.n);
  std::vector<uint32_t> idx(s->d_idx.n);
  HIP_TRY(hipSetDevice(s->device));
  if (!P.empty()) HIP_TRY(hipMemcpy(P.data(), s->d_P.p, P.size() * 4, hipMemcpyDeviceToHost));
  if (!idx.empty()) HIP_TRY(hipMemcpy(idx.data(), s->d_idx.p, idx.size() * 4, hipMemcpyDeviceToHost));
  build_bvh(P.data(), idx.data(), nt, &s->bvh);
  if (s->bvh.depth > 64) return fail(PBRT_HIP_ERR_LIMIT, "counters: canonical BVH deeper than the 64-entry traversal stack");
  PairNodes pairs;
  std::string why;
  if (!make_pair_nodes(s->bvh, &pairs, &why)) return fail(PBRT_HIP_ERR_LIMIT, "counters: " + why);
  s->d_nodes.release();
  HIP_TRY(s->d_nodes.alloc(pairs.q.size()));
  HIP_TRY(s->d_order_exact.alloc(nt));
  HIP_TRY(s->d_tris_exact.alloc(kTriStride * (size_t)nt));
  if (!pairs.q.empty()) HIP_TRY(hipMemcpyAsync(s->d_nodes.p, pairs.q.data(), pairs.q.size() * 16, hipMemcpyHostToDevice, s->stream));
  if (nt) HIP_TRY(hipMemcpyAsync(s->d_order_exact.p, s->bvh.order.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(launch_pack_tris(s->d_P.p, s->d_idx.p, s->d_mat_id.p, s->d_order_exact.p, nt, s->dev.n_tris, s->d_spheres.p, s->d_tris_exact.p, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->dev.nodes = s->d_nodes.p;  // (the pre-canonical allocation was empty and has just been released: no stale pointer is kept)
  s->device_bytes += s->d_nodes.n * 16 + s->d_tris_exact.n * 16 + s->d_order_exact.n * 4;
  s->dev_exact = s->dev;
  s->dev_exact.nodes = s->d_nodes.p;
  s->dev_exact.tris = s->d_tris_exact.p;
  s->dev_exact.n_nodes = (uint32_t)s->bvh.nodes.size();
  s->dev_exact.root_ref = pairs.root_ref;
  for (int k = 0; k < 3; k++) { s->dev_exact.root_lo[k] = pairs.root_lo[k]; s->dev_exact.root_hi[k] = pairs.root_hi[k]; }
  s->canonical_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  s->canonical_ready = true;
  return PBRT_HIP_OK;
}

}  // namespace

extern "C" {

int pbrt_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *pbrt_hip_last_error(void) { return pbrt_hip::last_error_message(); }
const char *pbrt_hip_version(void) { return "pbrt_hip 0.5 (gfx950)"; }
#ifndef PBRT_HIP_BUILD_ID
#define PBRT_HIP_BUILD_ID "unknown"
#endif
const char *pbrt_hip_build_id(void) { return PBRT_HIP_BUILD_ID; }

int pbrt_hip_bvh_build_host(const float *P, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris, uint32_t *nodes,
                            uint32_t *order, uint32_t *n_nodes, uint32_t *depth) {
  try {
    if ((n_tris && (!P || !idx)) || !n_nodes || !depth) return fail(PBRT_HIP_ERR_INVALID, "bvh_build_host: null argument");
    for (size_t i = 0; i < 3 * (size_t)n_tris; i++)
      if (idx[i] >= n_verts) return fail(PBRT_HIP_ERR_INVALID, "bvh_build_host: vertex index out of range");
    if (first_non_finite_vertex(P, idx, n_tris) >= 0) return fail(PBRT_HIP_ERR_INVALID, "bvh_build_host: a vertex is not finite");
    Bvh b;
    build_bvh(P, idx, n_tris, &b);
    *n_nodes = (uint32_t)b.nodes.size();
    *depth = b.depth;
    if (nodes && !b.nodes.empty()) std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(BvhNode));
    if (order && !b.order.empty()) std::memcpy(order, b.order.data(), b.order.size() * 4);
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_quad_build_host(const float *P, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris, int split_leaves,
                             uint32_t *quads, uint32_t cap_nodes, uint32_t *n_quads, uint32_t *stack_need) {
  return pbrt_hip_quad_build_host_ex(P, n_verts, idx, n_tris, split_leaves, PBRT_HIP_TREE_DEFAULT, quads, cap_nodes, n_quads, stack_need,
                                     nullptr, nullptr, nullptr, nullptr);
}

int pbrt_hip_quad_build_host_ex(const float *P, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris, int split_leaves, uint32_t tree,
                                uint32_t *quads, uint32_t cap_nodes, uint32_t *n_quads, uint32_t *stack_need, uint32_t *order,
                                float *root_box, uint32_t *n_refs, float *exact_boxes) {
  try {
    if ((n_tris && (!P || !idx)) || !n_quads || !stack_need) return fail(PBRT_HIP_ERR_INVALID, "quad_build_host: null argument");
    if (tree != PBRT_HIP_TREE_SAH && tree != PBRT_HIP_TREE_REINSERT && tree != PBRT_HIP_TREE_DEFAULT) return fail(PBRT_HIP_ERR_INVALID, "quad_build_host: unknown tree");
    for (size_t i = 0; i < 3 * (size_t)n_tris; i++)
      if (idx[i] >= n_verts) return fail(PBRT_HIP_ERR_INVALID, "quad_build_host: vertex index out of range");
    if (first_non_finite_vertex(P, idx, n_tris) >= 0) return fail(PBRT_HIP_ERR_INVALID, "quad_build_host: a vertex is not finite");
    Bvh b;
    build_bvh(P, idx, n_tris, &b);
    QuadNodes q;
    const ProductionTree pt = tree == PBRT_HIP_TREE_DEFAULT ? production_tree_default() : (ProductionTree)tree;
    build_production_quads(b, P, idx, n_tris, pt, split_leaves != 0, &q, n_refs);
    *n_quads = (uint32_t)(q.q.size() / 4);
    *stack_need = q.stack_need;
    if (order && !b.order.empty()) std::memcpy(order, b.order.data(), b.order.size() * 4);
    if (root_box && !b.nodes.empty())
      for (int a = 0; a < 3; a++) { root_box[a] = b.nodes[0].lo[a]; root_box[3 + a] = b.nodes[0].hi[a]; }
    if (quads) {
      if (*n_quads > cap_nodes) return fail(PBRT_HIP_ERR_LIMIT, "quad_build_host: output too small");
      if (!q.q.empty()) std::memcpy(quads, q.q.data(), q.q.size() * 16);
      if (exact_boxes && !q.exact.empty()) std::memcpy(exact_boxes, q.exact.data(), q.exact.size() * 4);
    }
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_scene_create(const pbrt_hip_scene_desc *d, int device, pbrt_hip_scene **out) {
  return pbrt_hip_scene_create_ex(d, device, 0u, out);  // the default: built and optimised on the device
}

int pbrt_hip_scene_create_ex(const pbrt_hip_scene_desc *d, int device, uint32_t flags, pbrt_hip_scene **out) {
  if (!d || !out) return fail(PBRT_HIP_ERR_INVALID, "scene_create: null argument");
  *out = nullptr;
  try {
    if (d->xres <= 0 || d->yres <= 0) return fail(PBRT_HIP_ERR_INVALID, "scene_create: resolution must be positive");
    if (d->n_tris && (!d->P || !d->idx || !d->mat_id)) return fail(PBRT_HIP_ERR_INVALID, "scene_create: missing mesh arrays");
    if ((d->n_tris || d->n_spheres) && (!d->mats || d->n_mats == 0)) return fail(PBRT_HIP_ERR_INVALID, "scene_create: no materials");
    if (d->n_mats > 65536) return fail(PBRT_HIP_ERR_LIMIT, "scene_create: more than 65536 materials");
    if (d->n_tris > (1u << 24)) return fail(PBRT_HIP_ERR_LIMIT, "scene_create: more than 2^24 triangles (leaf references hold a 24-bit slot)");
    if ((uint64_t)d->n_tris + d->n_spheres > (1u << 24)) return fail(PBRT_HIP_ERR_LIMIT, "scene_create: more than 2^24 primitives (triangles + spheres; leaf references hold a 24-bit slot)");
    if (d->n_spheres && !d->spheres) return fail(PBRT_HIP_ERR_INVALID, "scene_create: n_spheres > 0 but no sphere table");
    for (size_t i = 0; i < 3 * (size_t)d->n_tris; i++)
      if (d->idx[i] >= d->n_verts) return fail(PBRT_HIP_ERR_INVALID, "scene_create: vertex index out of range");
    for (uint32_t t = 0; t < d->n_tris; t++)
      if (d->mat_id[t] >= d->n_mats) return fail(PBRT_HIP_ERR_INVALID, "scene_create: material id out of range");
    {
      const long long bad = first_non_finite_vertex(d->P, d->idx, d->n_tris);
      if (bad >= 0) return fail(PBRT_HIP_ERR_INVALID, "scene_create: vertex " + std::to_string(bad) + " is not finite");
    }
    for (uint32_t s = 0; s < d->n_spheres; s++) {
      const pbrt_hip_sphere &sp = d->spheres[s];
      if (!std::isfinite(sp.c[0]) || !std::isfinite(sp.c[1]) || !std::isfinite(sp.c[2]) || !std::isfinite(sp.r) || !(sp.r > 0.f))
        return fail(PBRT_HIP_ERR_INVALID, "scene_create: sphere centre / radius must be finite and the radius positive");
    }
    // (a NaN in a light's position or in a colour travels into ray directions and throughputs: a ray that is not a number is pruned by
    // nothing and walks the whole tree -- minutes per frame on a large scene -- before its sample is dropped as NaN)
    if (d->n_lights && !d->lights) return fail(PBRT_HIP_ERR_INVALID, "scene_create: n_lights > 0 but no light table");
    for (uint32_t i = 0; i < d->n_lights; i++)
      for (int k = 0; k < 3; k++)
        if (!std::isfinite(d->lights[i].p[k]) || !std::isfinite(d->lights[i].c[k]))
          return fail(PBRT_HIP_ERR_INVALID, "scene_create: light " + std::to_string(i) + ": position / direction / colour is not finite");
    for (uint32_t i = 0; i < d->n_mats; i++)
      for (int k = 0; k < 3; k++)
        if (!std::isfinite(d->mats[i].k[k]) || !std::isfinite(d->mats[i].le[k]))
          return fail(PBRT_HIP_ERR_INVALID, "scene_create: material " + std::to_string(i) + ": colour / emission is not finite");
    for (int k = 0; k < 16; k++)
      if (!std::isfinite(d->cam_to_world[k])) return fail(PBRT_HIP_ERR_INVALID, "scene_create: camera matrix is not finite");
    if (!(d->fov > 0.f && d->fov < 180.f)) return fail(PBRT_HIP_ERR_INVALID, "scene_create: fov must lie in (0, 180) degrees");
    for (int k = 0; k < 4; k++)  // Film "cropwindow": fractions of the film (film.rs:92-101 multiplies and rounds them up: a NaN or 1e30 there is an int overflow)
      if (!(d->crop[k] >= 0.f && d->crop[k] <= 1.f)) return fail(PBRT_HIP_ERR_INVALID, "scene_create: crop window values must lie in [0, 1]");
    for (uint32_t s = 0; s < d->n_spheres; s++)
      if (d->spheres[s].mat >= d->n_mats) return fail(PBRT_HIP_ERR_INVALID, "scene_create: sphere material id out of range");
    bool textured = false;      // a triangle whose material's Kd is a texture: its corner (u, v) must be there
    bool textured_sph = false;  // a sphere whose material's Kd is one: (u, v) from its own parametrisation (kernels.hip sphere_uv)
    if (d->n_textures && !d->textures) return fail(PBRT_HIP_ERR_INVALID, "scene_create: n_textures > 0 but no texture table");
    for (uint32_t i = 0; i < d->n_mats; i++) {
      if (d->mats[i].kd_tex > d->n_textures) return fail(PBRT_HIP_ERR_INVALID, "scene_create: material texture number out of range");
      if (d->mats[i].kd_tex && !d->textures) return fail(PBRT_HIP_ERR_INVALID, "scene_create: textured material but no texture table");
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
      const pbrt_hip_texture &tx = d->textures[i];
      if (tx.type != 0u) return fail(PBRT_HIP_ERR_INVALID, "scene_create: unknown texture type");
      if (!std::isfinite(tx.su) || !std::isfinite(tx.sv) || !std::isfinite(tx.du) || !std::isfinite(tx.dv))
        return fail(PBRT_HIP_ERR_INVALID, "scene_create: texture mapping is not finite");
      for (int k = 0; k < 3; k++)
        if (!std::isfinite(tx.tex1[k]) || !std::isfinite(tx.tex2[k])) return fail(PBRT_HIP_ERR_INVALID, "scene_create: texture colour is not finite");
    }
    for (uint32_t t = 0; t < d->n_tris && !textured; t++) textured = d->mats[d->mat_id[t]].kd_tex != 0u && d->mats[d->mat_id[t]].type == 0u;
    for (uint32_t i = 0; i < d->n_spheres && !textured_sph; i++) textured_sph = d->mats[d->spheres[i].mat].kd_tex != 0u && d->mats[d->spheres[i].mat].type == 0u;
    if (textured && !d->tri_uv) return fail(PBRT_HIP_ERR_INVALID, "scene_create: a triangle's material is textured but tri_uv is NULL");
    if (textured)
      for (size_t i = 0; i < 6 * (size_t)d->n_tris; i++)
        if (!std::isfinite(d->tri_uv[i])) return fail(PBRT_HIP_ERR_INVALID, "scene_create: tri_uv is not finite");

    int ndev = pbrt_hip_device_count();
    if (ndev <= 0) return fail(PBRT_HIP_ERR_NO_DEVICE, "scene_create: no HIP device (there is no CPU fallback)");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(PBRT_HIP_ERR_INVALID, "scene_create: device index out of range");
    HIP_TRY(hipSetDevice(device));

    std::unique_ptr<pbrt_hip_scene> s(new pbrt_hip_scene());
    s->device = device;
    {
      int cus = 0;
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
      s->n_cu = cus > 0 ? (uint32_t)cus : 1u;
    }
    s->desc = *d;
    s->desc.P = nullptr; s->desc.idx = nullptr; s->desc.mat_id = nullptr;
    s->desc.mats = nullptr; s->desc.lights = nullptr; s->desc.spheres = nullptr;
    s->desc.tri_uv = nullptr; s->desc.textures = nullptr;
    s->textured = textured || textured_sph;

    // --- accelerator: ONE default -- the device builder further down (binned SAH + parallel re-insertion + collapse), whoever asks
    // and however (pbrt_hip_scene_create, flags 0, pbrt_hip_render_multi, the command line, bench.py); the host's binned-SAH
    // builder only on request (PBRT_HIP_SCENE_HOST_BUILD / _OPTIMIZED_TREE, or PBRT_HIP_BUILDER=host in the environment when the
    // caller left the choice open) ---
    if (flags & ~(PBRT_HIP_SCENE_GPU_BUILD | PBRT_HIP_SCENE_OPTIMIZED_TREE | PBRT_HIP_SCENE_PLAIN_TREE | PBRT_HIP_SCENE_HOST_BUILD))
      return fail(PBRT_HIP_ERR_INVALID, "scene_create: unknown flag");
    const bool want_host = (flags & (PBRT_HIP_SCENE_HOST_BUILD | PBRT_HIP_SCENE_OPTIMIZED_TREE)) != 0u;
    const bool want_gpu = (flags & (PBRT_HIP_SCENE_GPU_BUILD | PBRT_HIP_SCENE_PLAIN_TREE)) != 0u;
    if (want_host && want_gpu)
      return fail(PBRT_HIP_ERR_INVALID, "scene_create: PBRT_HIP_SCENE_HOST_BUILD / _OPTIMIZED_TREE are host builds, not combined with PBRT_HIP_SCENE_GPU_BUILD / _PLAIN_TREE");
    if (!want_host && !want_gpu) {
      const char *b = std::getenv("PBRT_HIP_BUILDER");
      if (b && std::strcmp(b, "host") == 0) flags |= PBRT_HIP_SCENE_HOST_BUILD;
      else flags |= PBRT_HIP_SCENE_GPU_BUILD;
    } else if (want_gpu) {
      flags |= PBRT_HIP_SCENE_GPU_BUILD;  // (PBRT_HIP_SCENE_PLAIN_TREE alone qualifies the default)
    }
    // Spheres are PRIMITIVES OF THE TREE (round 6; until round 5 every ray tested every sphere after the walk).  Every builder here --
    // the host's binned SAH, the device builder, the collapse, the lazily built canonical tree -- bounds a primitive by the box of its
    // three vertices, so sphere s enters the vertex / index buffers as a degenerate PROXY TRIANGLE (c - r, c + r, c - r): primitive
    // n_tris + s, bounded by exactly the sphere's box [c - r, c + r] (fp32 per component: the oracle's sphere_box), centroid its centre.
    // Its leaf record is a sphere's (pack_tris_kernel) and the leaf pass runs the sphere test on it (trav_run<..., SPH>).
    const uint32_t np = d->n_tris + d->n_spheres;
    std::vector<float> P_aug;
    std::vector<uint32_t> idx_aug;
    std::vector<uint16_t> mat_aug;
    const float *bP = d->P;
    const uint32_t *bidx = d->idx;
    const uint16_t *bmat = d->mat_id;
    uint32_t n_verts_b = d->n_verts;
    if (d->n_spheres) {
      P_aug.assign(d->P, d->P + (d->n_tris ? 3 * (size_t)d->n_verts : 0));
      if (!d->n_tris) n_verts_b = 0;
      idx_aug.assign(d->idx, d->idx + 3 * (size_t)d->n_tris);
      mat_aug.assign(d->mat_id, d->mat_id + d->n_tris);
      for (uint32_t i = 0; i < d->n_spheres; i++) {
        const pbrt_hip_sphere &sp = d->spheres[i];
        const uint32_t v0 = n_verts_b + 2 * i;
        for (int k = 0; k < 3; k++) P_aug.push_back(sp.c[k] - sp.r);
        for (int k = 0; k < 3; k++) P_aug.push_back(sp.c[k] + sp.r);
        idx_aug.push_back(v0); idx_aug.push_back(v0 + 1); idx_aug.push_back(v0);
        mat_aug.push_back((uint16_t)sp.mat);
        if (!std::isfinite(sp.c[0] - sp.r) || !std::isfinite(sp.c[0] + sp.r) || !std::isfinite(sp.c[1] - sp.r) || !std::isfinite(sp.c[1] + sp.r) ||
            !std::isfinite(sp.c[2] - sp.r) || !std::isfinite(sp.c[2] + sp.r))
          return fail(PBRT_HIP_ERR_INVALID, "scene_create: a sphere's bounding box is not finite");
      }
      n_verts_b += 2 * d->n_spheres;
      bP = P_aug.data(); bidx = idx_aug.data(); bmat = mat_aug.data();
    }
    s->gpu_built = (flags & PBRT_HIP_SCENE_GPU_BUILD) && np >= 2;
    PairNodes pairs;
    if (!s->gpu_built) {
      const auto t0 = std::chrono::steady_clock::now();
      build_bvh(bP, bidx, np, &s->bvh);
      if (s->bvh.depth > 64) return fail(PBRT_HIP_ERR_LIMIT, "scene_create: BVH deeper than the 64-entry traversal stack");
      std::string why;
      if (!make_pair_nodes(s->bvh, &pairs, &why)) return fail(PBRT_HIP_ERR_LIMIT, "scene_create: " + why);
      s->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    // --- light table: explicit lights, then every emissive triangle in index order ---
    std::vector<float4> lights;
    float le_inf[3] = {0.f, 0.f, 0.f};
    bool has_inf = false;
    auto as_f = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    for (uint32_t i = 0; i < d->n_lights; i++) {
      const pbrt_hip_light &l = d->lights[i];
      if (l.type > 2) return fail(PBRT_HIP_ERR_INVALID, "scene_create: unknown light type");
      lights.push_back(make_float4(as_f(l.type), l.p[0], l.p[1], l.p[2]));
      lights.push_back(make_float4(0, 0, 0, 0));
      lights.push_back(make_float4(0, 0, 0, 0));
      lights.push_back(make_float4(l.c[0], l.c[1], l.c[2], 0));
      lights.push_back(make_float4(0, 0, 0, 0));
      if (l.type == 2) {
        for (int k = 0; k < 3; k++) le_inf[k] = le_inf[k] + l.c[k];
        has_inf = true;
      }
    }
    for (uint32_t t = 0; t < d->n_tris; t++) {
      const pbrt_hip_material &m = d->mats[d->mat_id[t]];
      if (!(m.le[0] > 0.f || m.le[1] > 0.f || m.le[2] > 0.f)) continue;
      F3 p[3];
      for (int v = 0; v < 3; v++) {
        const float *q = d->P + 3 * (size_t)d->idx[3 * (size_t)t + v];
        p[v] = {q[0], q[1], q[2]};
      }
      F3 cr = cross3(sub(p[1], p[0]), sub(p[2], p[0]));
      float len = std::sqrt(dot3(cr, cr));
      lights.push_back(make_float4(as_f(3u), p[0].x, p[0].y, p[0].z));
      lights.push_back(make_float4(p[1].x, p[1].y, p[1].z, 0.5f * len));
      lights.push_back(make_float4(p[2].x, p[2].y, p[2].z, 0));
      lights.push_back(make_float4(m.le[0], m.le[1], m.le[2], 0));
      lights.push_back(make_float4(cr.x / len, cr.y / len, cr.z / len, 0));
    }
    s->n_lights = (uint32_t)(lights.size() / 5);

    std::vector<float4> mats(2 * (size_t)d->n_mats);
    for (uint32_t i = 0; i < d->n_mats; i++) {
      const pbrt_hip_material &m = d->mats[i];
      if (m.type > 1) return fail(PBRT_HIP_ERR_INVALID, "scene_create: unknown material type");
      mats[2 * i] = make_float4(as_f(m.type), m.k[0], m.k[1], m.k[2]);
      mats[2 * i + 1] = make_float4(m.le[0], m.le[1], m.le[2], as_f(m.type == 0u ? m.kd_tex : 0u));
    }
    std::vector<float4> spheres(2 * (size_t)d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++) {
      const pbrt_hip_sphere &sp = d->spheres[i];
      spheres[2 * i] = make_float4(sp.c[0], sp.c[1], sp.c[2], sp.r);
      spheres[2 * i + 1] = make_float4(as_f(sp.mat), 0, 0, 0);
    }

    // --- upload: vertex / index buffers, flattened nodes, leaf order; pack leaf records on device ---
    const uint32_t nt = np;  // primitives: the triangles + the spheres' proxies (D.n_tris below stays the TRIANGLES: primitive ids >= it are spheres)
    HIP_TRY(s->d_P.alloc(3 * (size_t)n_verts_b));
    HIP_TRY(s->d_idx.alloc(3 * (size_t)nt));
    HIP_TRY(s->d_mat_id.alloc(nt));
    HIP_TRY(s->d_order.alloc(nt));
    QuadNodes quads;
    if (!s->gpu_built) {
      const auto t0 = std::chrono::steady_clock::now();
      const char *sl = debug_knob("PBRT_HIP_SPLIT_LEAVES");
      build_production_quads(s->bvh, bP, bidx, np, (flags & PBRT_HIP_SCENE_OPTIMIZED_TREE) ? kTreeReinsert : production_tree_default(),
                             !(sl && sl[0] == '0'), &quads);
      s->build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    HIP_TRY(s->d_nodes.alloc(pairs.q.size()));
    HIP_TRY(s->d_quads.alloc(s->gpu_built ? 4 * (size_t)nt : quads.q.size()));
    HIP_TRY(s->d_tris.alloc(kTriStride * (size_t)nt));
    HIP_TRY(s->d_mats.alloc(mats.size()));
    HIP_TRY(s->d_lights.alloc(lights.size()));
    HIP_TRY(s->d_spheres.alloc(spheres.size()));
    HIP_TRY(s->d_counters.alloc(80));  // [0..4] ray / visit counters, [6..7] pixel-order scratch, [8..71] the pixel hand-out counters
    HIP_TRY(hipStreamCreate(&s->stream));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    auto up = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
      if (!bytes) return hipSuccess;
      return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream);
    };
    HIP_TRY(up(s->d_P.p, bP, s->d_P.n * 4));
    HIP_TRY(up(s->d_idx.p, bidx, s->d_idx.n * 4));
    HIP_TRY(up(s->d_mat_id.p, bmat, s->d_mat_id.n * 2));
    GpuBuildInfo gb{};
    if (s->gpu_built) {
      HIP_TRY(gpu_build_quads(s->d_P.p, s->d_idx.p, nt, s->d_order.p, s->d_quads.p, nt, (flags & PBRT_HIP_SCENE_PLAIN_TREE) ? 0u : kGpuBuildReinsert, &gb,
                              s->stream));
      if (gb.stack_need + 1u > 4096u) return fail(PBRT_HIP_ERR_LIMIT, "scene_create: device-built tree too deep");
      quads.stack_need = gb.stack_need;
      s->build_ms = gb.build_ms;
      s->reinsert_passes = gb.reinsert_passes;
      s->reinsert_moves = gb.reinsert_moves;
      s->reinsert_ms = gb.reinsert_ms;
      s->reinsert_cost_before = gb.reinsert_cost_before;
      s->reinsert_cost_after = gb.reinsert_cost_after;
      s->reinsert_undone = gb.reinsert_undone;
    } else {
      HIP_TRY(up(s->d_order.p, s->bvh.order.data(), s->d_order.n * 4));
    }
    HIP_TRY(up(s->d_nodes.p, pairs.q.data(), pairs.q.size() * 16));
    if (!s->gpu_built) HIP_TRY(up(s->d_quads.p, quads.q.data(), quads.q.size() * 16));
    HIP_TRY(up(s->d_mats.p, mats.data(), mats.size() * 16));
    HIP_TRY(up(s->d_lights.p, lights.data(), lights.size() * 16));
    HIP_TRY(up(s->d_spheres.p, spheres.data(), spheres.size() * 16));
    HIP_TRY(launch_pack_tris(s->d_P.p, s->d_idx.p, s->d_mat_id.p, s->d_order.p, nt, d->n_tris, s->d_spheres.p, s->d_tris.p, s->stream));
    if (textured || textured_sph) {  // corner (u, v) into leaf-slot order (whichever builder made d_order), the texture table as 3 x 16 B records
      std::vector<float4> tex(3 * (size_t)d->n_textures);
      for (uint32_t i = 0; i < d->n_textures; i++) {
        const pbrt_hip_texture &tx = d->textures[i];
        tex[3 * i] = make_float4(as_f(tx.type), tx.tex1[0], tx.tex1[1], tx.tex1[2]);
        tex[3 * i + 1] = make_float4(tx.tex2[0], tx.tex2[1], tx.tex2[2], tx.su);
        tex[3 * i + 2] = make_float4(tx.sv, tx.du, tx.dv, 0.f);
      }
      HIP_TRY(s->d_textures.alloc(tex.size()));
      HIP_TRY(up(s->d_textures.p, tex.data(), tex.size() * 16));
      if (textured) {
        HIP_TRY(s->d_tri_uv_in.alloc(6 * (size_t)d->n_tris));
        HIP_TRY(s->d_tri_uv.alloc(3 * (size_t)nt));
        HIP_TRY(up(s->d_tri_uv_in.p, d->tri_uv, 24 * (size_t)d->n_tris));
        HIP_TRY(launch_pack_uv(s->d_tri_uv_in.p, s->d_order.p, nt, d->n_tris, s->d_tri_uv.p, s->stream));
      }
      HIP_TRY(hipStreamSynchronize(s->stream));  // (tex is a local)
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->device_bytes = s->d_tri_uv_in.n * 4 + s->d_tri_uv.n * 8 + s->d_textures.n * 16 + s->d_P.n * 4 + s->d_idx.n * 4 + s->d_mat_id.n * 2 + s->d_order.n * 4 + s->d_nodes.n * 16 + s->d_quads.n * 16 +
                      s->d_tris.n * 16 + s->d_mats.n * 16 + s->d_lights.n * 16 + s->d_spheres.n * 16;

    // --- kernel argument block ---
    DevScene &D = s->dev;
    D.nodes = s->d_nodes.p;
    D.quads = s->d_quads.p;
    D.quad_stack_need = quads.stack_need;
    D.tris = s->d_tris.p;
    D.mats = s->d_mats.p;
    D.lights = s->d_lights.p;
    D.spheres = s->d_spheres.p;
    D.n_nodes = (uint32_t)s->bvh.nodes.size();
    D.root_ref = pairs.root_ref;
    for (int k = 0; k < 3; k++) { D.root_lo[k] = pairs.root_lo[k]; D.root_hi[k] = pairs.root_hi[k]; }
    if (s->gpu_built) {  // the walk enters quad 0 through the root box; there is no canonical binary tree
      D.n_nodes = 2u * nt - 1u;
      D.root_ref = 0u;
      for (int k = 0; k < 3; k++) { D.root_lo[k] = gb.root_lo[k]; D.root_hi[k] = gb.root_hi[k]; }
      s->n_quads_gpu = gb.n_quads;
    }
    D.inv_parallel = inv_parallel_for_extent(std::max(D.root_hi[0] - D.root_lo[0], std::max(D.root_hi[1] - D.root_lo[1], D.root_hi[2] - D.root_lo[2])));
    D.n_tris = d->n_tris;  // (the triangles: a hit's primitive id >= this is sphere id - n_tris)
    s->n_prims = nt;
    D.n_spheres = d->n_spheres;
    D.n_lights = s->n_lights;
    D.n_lights_f = (float)s->n_lights;
    for (int k = 0; k < 3; k++) D.le_inf[k] = le_inf[k];
    D.has_inf = has_inf ? 1u : 0u;
    for (int k = 0; k < 12; k++) D.c2w[k] = d->cam_to_world[k];
    // perspective camera: screen window from the aspect ratio, fov on the shorter axis
    const float aspect = (float)d->xres / (float)d->yres;
    float sx0, sx1, sy0, sy1;
    if (aspect > 1.f) { sx0 = -aspect; sx1 = aspect; sy0 = -1.f; sy1 = 1.f; }
    else { sx0 = -1.f; sx1 = 1.f; sy0 = -1.f / aspect; sy1 = 1.f / aspect; }
    const float tan_half = (float)std::tan((double)d->fov * (3.14159265358979323846 / 180.0) * 0.5);
    D.cam_ax = ((sx1 - sx0) / (float)d->xres) * tan_half;
    D.cam_bx = sx0 * tan_half;
    D.cam_ay = -((sy1 - sy0) / (float)d->yres) * tan_half;
    D.cam_by = sy1 * tan_half;
    D.xres = d->xres;
    D.yres = d->yres;
    int32_t cb[4];
    film_cropped_bounds(d->xres, d->yres, d->crop, cb);
    D.cx0 = cb[0]; D.cy0 = cb[1]; D.cx1 = cb[2]; D.cy1 = cb[3];
    *out = s.release();
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

void pbrt_hip_scene_destroy(pbrt_hip_scene *scene) {
  if (!scene) return;
  (void)hipSetDevice(scene->device);
  delete scene;
}

int pbrt_hip_scene_info(const pbrt_hip_scene *s, uint32_t *n_nodes, uint32_t *depth, uint32_t *n_lights,
                        uint64_t *device_bytes) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "scene_info: null scene");
  if (n_nodes) *n_nodes = (uint32_t)s->bvh.nodes.size();
  if (depth) *depth = s->bvh.depth;
  if (n_lights) *n_lights = s->n_lights;
  if (device_bytes) *device_bytes = s->device_bytes;
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_walk_info(const pbrt_hip_scene *s, uint32_t *quad_nodes, uint32_t *stack_need) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "walk_info: null scene");
  if (quad_nodes) *quad_nodes = s->gpu_built ? s->n_quads_gpu : (uint32_t)(s->d_quads.n / 4);
  if (stack_need) *stack_need = s->dev.quad_stack_need;
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_build_info(const pbrt_hip_scene *s, uint32_t *gpu_built, double *build_ms) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "build_info: null scene");
  if (gpu_built) *gpu_built = s->gpu_built ? 1u : 0u;
  if (build_ms) *build_ms = s->build_ms;
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_optimize_info(const pbrt_hip_scene *s, uint32_t *passes, uint32_t *moves, double *ms) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "optimize_info: null scene");
  if (passes) *passes = s->reinsert_passes;
  if (moves) *moves = s->reinsert_moves;
  if (ms) *ms = s->reinsert_ms;
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_optimize_cost(const pbrt_hip_scene *s, double *before, double *after, uint32_t *undone) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "optimize_cost: null scene");
  if (before) *before = s->reinsert_cost_before;
  if (after) *after = s->reinsert_cost_after;
  if (undone) *undone = s->reinsert_undone;
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_canonical_info(const pbrt_hip_scene *s, uint32_t *ready, double *build_ms) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "canonical_info: null scene");
  if (ready) *ready = (s->canonical_ready || !s->gpu_built) ? 1u : 0u;
  if (build_ms) *build_ms = s->gpu_built ? s->canonical_build_ms : s->build_ms;
  return PBRT_HIP_OK;
}

int pbrt_hip_render_stack_plan(uint32_t stack_need, uint32_t *lds_rows, uint32_t *waves_per_cu, uint32_t *overflow_entries) {
  const RenderStackPlan p = render_stack_plan(stack_need, render_force_overflow(), render_prefer_lds());
  if (lds_rows) *lds_rows = p.rows;
  if (waves_per_cu) *waves_per_cu = p.waves_per_cu;
  if (overflow_entries) *overflow_entries = p.extra_entries;
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_export_quads(const pbrt_hip_scene *s, uint32_t *quads, uint32_t cap_nodes, uint32_t *n_quads, uint32_t *order) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "export_quads: null scene");
  const uint32_t n = s->gpu_built ? s->n_quads_gpu : (uint32_t)(s->d_quads.n / 4);
  if (n_quads) *n_quads = n;
  HIP_TRY(hipSetDevice(s->device));
  if (quads) {
    if (cap_nodes < n) return fail(PBRT_HIP_ERR_LIMIT, "export_quads: output too small");
    if (n) HIP_TRY(hipMemcpy(quads, s->d_quads.p, 64 * (size_t)n, hipMemcpyDeviceToHost));
  }
  if (order && s->d_order.n) HIP_TRY(hipMemcpy(order, s->d_order.p, 4 * s->d_order.n, hipMemcpyDeviceToHost));
  return PBRT_HIP_OK;
}

int pbrt_hip_scene_export_bvh(const pbrt_hip_scene *s, uint32_t *nodes, uint32_t *order) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "export_bvh: null scene");
  if (nodes && !s->bvh.nodes.empty()) std::memcpy(nodes, s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(BvhNode));
  if (order && !s->bvh.order.empty()) std::memcpy(order, s->bvh.order.data(), s->bvh.order.size() * 4);
  return PBRT_HIP_OK;
}

static int check_render_desc(const pbrt_hip_scene *s, const pbrt_hip_render_desc *r) {
  if (!s || !r) return fail(PBRT_HIP_ERR_INVALID, "render: null argument");
  if (r->spp_x == 0 || r->spp_y == 0) return fail(PBRT_HIP_ERR_INVALID, "render: spp_x and spp_y must be >= 1");
  if (r->world_size == 0 || r->rank >= r->world_size) return fail(PBRT_HIP_ERR_INVALID, "render: rank must be < world_size");
  if (r->integrator > PBRT_HIP_INTEGRATOR_PATH_MIS) return fail(PBRT_HIP_ERR_INVALID, "render: unknown integrator");
  if (r->sampler > PBRT_HIP_SAMPLER_HALTON) return fail(PBRT_HIP_ERR_INVALID, "render: unknown sampler");
  // (samplers 2 and 3 -- Sobol' proper and Halton -- share one instantiation of the kernel: "the table samplers")
  const bool table_sampler = r->sampler == PBRT_HIP_SAMPLER_SOBOL_ND || r->sampler == PBRT_HIP_SAMPLER_HALTON;
  if (table_sampler && (r->flags & (PBRT_HIP_FLAG_COUNTERS | PBRT_HIP_FLAG_WALK_COUNTERS)))
    return fail(PBRT_HIP_ERR_INVALID, "render: the counter flags are not available with the Sobol' / Halton samplers (samplers 2, 3)");
  // the kernels pack the sample index into 20 bits and the bounce count into 10 (kernels.hip path_store): beyond that a
  // persistent wave would never see its pixel finish
  if ((uint64_t)r->spp_x * (uint64_t)r->spp_y > PBRT_HIP_MAX_SPP)
    return fail(PBRT_HIP_ERR_LIMIT, "render: more than 2^20 samples per pixel");
  if (r->max_depth > PBRT_HIP_MAX_DEPTH) return fail(PBRT_HIP_ERR_LIMIT, "render: maxdepth above 1023");
  // box filter radii, box.rs:57-61 (0 = the default 0.5): any positive radius up to 16 pixels
  const float fx = filter_radius(r->filter_xwidth), fy = filter_radius(r->filter_ywidth);
  if (!(fx > 0.f) || !(fy > 0.f) || !std::isfinite(fx) || !std::isfinite(fy)) return fail(PBRT_HIP_ERR_INVALID, "render: the filter radii must be positive");
  if (fx > 16.f || fy > 16.f) return fail(PBRT_HIP_ERR_LIMIT, "render: filter radius above 16 pixels");
  if ((fx != 0.5f || fy != 0.5f) && (r->flags & (PBRT_HIP_FLAG_COUNTERS | PBRT_HIP_FLAG_WALK_COUNTERS)))
    return fail(PBRT_HIP_ERR_INVALID, "render: the counter flags need the default box filter (radius 0.5)");
  // (textures, the MIS integrator, the table samplers and a wide box filter combine freely -- render_kernel_x --; only the counting
  // instantiations exist for the default path alone)
  if ((s->textured || r->integrator == PBRT_HIP_INTEGRATOR_PATH_MIS) && (r->flags & (PBRT_HIP_FLAG_COUNTERS | PBRT_HIP_FLAG_WALK_COUNTERS)))
    return fail(PBRT_HIP_ERR_LIMIT, "render: the counter flags are not available for textured materials / the MIS integrator");
  if (!(r->max_sample_luminance >= 0.f)) return fail(PBRT_HIP_ERR_INVALID, "render: max_sample_luminance must be >= 0 (0 = none)");
  if (fx != 0.5f || fy != 0.5f) {
    // the fixed-point film: a sample adds at most 2^39 units to a pixel's int64 accumulator, and a pixel receives
    // at most spp x footprint samples (from every rank together: the N-rank reduce adds the same samples) -- 2^24 of them fit
    const uint64_t foot = (uint64_t)(2 * (int)std::ceil(fx) + 1) * (uint64_t)(2 * (int)std::ceil(fy) + 1);
    if ((uint64_t)r->spp_x * (uint64_t)r->spp_y * foot > (1ull << 24))
      return fail(PBRT_HIP_ERR_LIMIT, "render: samples per pixel x filter footprint above 2^24 (the fixed-point film's accumulators could wrap)");
  }
  return PBRT_HIP_OK;
}

namespace {
// The scratch one render of `s` needs beyond the caller's slab, sized for THIS description and (re)allocated here when what the
// scene holds is too small: the lanes' path-state records, the partial film sums of the work items, the overflow area of the
// walk's stack, the generator matrices of sampler 2.  pbrt_hip_render_device calls it; a host that is about to launch on several
// GPUs calls it for every GPU FIRST (pbrt_hip_render_prepare), so that no hipMalloc -- a synchronising call -- sits between
// the launches of a frame.
struct RenderScratch {
  uint32_t n_workgroups = 0, chunk_shift = 0;
  uint32_t passes = 1;  // partials_passes
  RenderStackPlan plan{};
};
// The partial film sums cost 16 K bytes per pixel of the rank's share (one float4 per item, K <= 16 chunks per pixel): 1.07 GB for C3, 4.3 GB
// for C4's 4096^2 on one GPU, and growing with the resolution.  A frame whose sums would pass the cap (2 GiB; PBRT_HIP_PARTIALS_CAP_KB for the
// tests) is rendered in P passes over the same buffer: pass p takes the rank's super-tiles j = p + P * j', which is exactly the share of rank
// `rank + world * p` of `world * P` ranks -- so the render kernel runs unchanged -- and the merge puts tile j' of the pass at tile j of the
// rank's slab.  Every pixel keeps its samples, chunks and order of additions: the film is the one-pass film bit for bit.  Passes also keep
// the item numbers of a launch inside 32 bits.
uint32_t partials_passes(uint32_t n_local, uint32_t n_chunks) {
  if (n_local == 0) return 1;
  const uint64_t cap_bytes = (uint64_t)std::max<uint32_t>(1u, tuning("PBRT_HIP_PARTIALS_CAP_KB", 2u << 20, 1l << 30)) << 10;
  const uint64_t per_tile = 4096ull * n_chunks * 16ull;
  uint64_t tiles = std::max<uint64_t>(1, cap_bytes / per_tile);
  tiles = std::min<uint64_t>(tiles, ((1ull << 32) - 1) / (4096ull * n_chunks));
  return (uint32_t)((n_local + tiles - 1) / tiles);
}
int ensure_render_scratch(pbrt_hip_scene *s, const pbrt_hip_render_desc *r, const FilmGeom &fg, const Shard &sh, RenderScratch *out) {
  const uint32_t spp = r->spp_x * r->spp_y;
  out->chunk_shift = sample_chunk_shift(spp);
  const uint32_t n_chunks = 1u << out->chunk_shift;  // K: DESIGN.md 3.1
  out->passes = fg.wide ? 1u : partials_passes(sh.n_local, n_chunks);
  const uint32_t n_pass_tiles = (sh.n_local + out->passes - 1) / out->passes;  // of pass 0, the largest
  if ((uint64_t)n_pass_tiles * 4096u * n_chunks >= (1ull << 32)) return fail(PBRT_HIP_ERR_LIMIT, "render: film too large for 32-bit item numbers");
  if ((uint64_t)r->world_size * out->passes >= (1ull << 32)) return fail(PBRT_HIP_ERR_LIMIT, "render: world_size x passes does not fit 32 bits");
  // (scenes with spheres run a kernel with a bigger register budget, 3 waves per SIMD: kernels.hip)
  out->plan = render_stack_plan(s->dev.quad_stack_need, render_force_overflow(), render_prefer_lds());
  // (the instantiations for another filter radius and for the Sobol' sampler fit the 96 VGPRs of 5 waves per SIMD like the default one)
  const uint32_t waves_per_cu = s->dev.n_spheres ? std::min(kRenderWavesPerCuSpheres, out->plan.waves_per_cu) : out->plan.waves_per_cu;
  out->n_workgroups = std::min<uint32_t>(n_pass_tiles * 64u * n_chunks, std::max<uint32_t>(1u, tuning("PBRT_HIP_RENDER_WORKGROUPS", s->n_cu * waves_per_cu, 1 << 20)));
  if (r->sampler == PBRT_HIP_SAMPLER_SOBOL_ND && s->d_sobol.n == 0) {
    static_assert(kSobolNdDims == 2 * (int)kSobolNdRequests, "sampler 2: two dimensions per request");
    std::vector<uint32_t> mat((size_t)kSobolNdDims * 32);
    sobol_nd_matrices(mat.data());
    HIP_TRY(s->d_sobol.alloc(mat.size()));
    HIP_TRY(hipMemcpy(s->d_sobol.p, mat.data(), mat.size() * 4, hipMemcpyHostToDevice));
  }
  if (r->sampler == PBRT_HIP_SAMPLER_HALTON && s->d_halton.n == 0) {
    static_assert(kHaltonDims == 2 * (int)kSobolNdRequests, "sampler 3: two dimensions per request, as many requests as sampler 2");
    uint32_t tab[kHaltonDims * 4];
    halton_table(tab);
    HIP_TRY(s->d_halton.alloc(kHaltonDims * 4));
    HIP_TRY(hipMemcpy(s->d_halton.p, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  {
    // float4 records: 5 x 64 per one-wave workgroup (kernels.hip LaneRecords); with another box filter radius 16 x 2 x 64 more
    // behind them (kWideSlotFloat4: a chunk's sums per footprint)
    const size_t path = (size_t)out->n_workgroups * 320, need = path + (fg.wide ? (size_t)out->n_workgroups * 2048 : 0);
    if (s->d_lane_state.n < need) { s->d_lane_state.release(); HIP_TRY(s->d_lane_state.alloc(need)); }
  }
  if (!fg.wide) {
    const size_t need = (size_t)n_pass_tiles * 4096u * n_chunks;  // one float4 per item of a pass
    // (C3 1.07 GB in one pass; C4's 4096^2 x 16 chunks on one GPU: 3 passes over 1.43 GB; the buffer follows the frame: released when a
    // later render needs less than a quarter of it)
    if (s->d_partials.n < need || s->d_partials.n / 4 > need) { s->d_partials.release(); HIP_TRY(s->d_partials.alloc(need)); }
  }
  {
    // the overflow variant keeps kQuadLdsStackOvf rows per lane in LDS; deeper entries (rare) go here
    const size_t need = (size_t)out->n_workgroups * 64 * out->plan.extra_entries;
    if (s->d_stack_overflow.n < need) { s->d_stack_overflow.release(); HIP_TRY(s->d_stack_overflow.alloc(need)); }
  }
  return PBRT_HIP_OK;
}
}  // namespace

int pbrt_hip_render_prepare(pbrt_hip_scene *s, const pbrt_hip_render_desc *r) {
  int rc = check_render_desc(s, r);
  if (rc) return rc;
  try {
    if (s->pending) return fail(PBRT_HIP_ERR_INVALID, "render_prepare: a render of this scene is still in flight (call pbrt_hip_render_wait first)");
    HIP_TRY(hipSetDevice(s->device));
    const FilmGeom fg = film_geom(s->desc, *r);
    const Shard sh = make_shard_bounds(fg.sb, r->rank, r->world_size);
    RenderScratch rs;
    return ensure_render_scratch(s, r, fg, sh, &rs);
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_render_device(pbrt_hip_scene *s, const pbrt_hip_render_desc *r, void *d_slab, void *stream) {
  int rc = check_render_desc(s, r);
  if (rc) return rc;
  try {
    // the events, counters and lane-state records of a scene serve one render at a time
    if (s->pending) return fail(PBRT_HIP_ERR_INVALID, "render_device: a render of this scene is still in flight (call pbrt_hip_render_wait first)");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    const FilmGeom fg = film_geom(s->desc, *r);
    const Shard sh = make_shard_bounds(fg.sb, r->rank, r->world_size);
    if ((sh.n_local || (fg.wide && fg.crop_px())) && !d_slab) return fail(PBRT_HIP_ERR_INVALID, "render_device: null slab");
    RenderParams R;
    R.sx0 = fg.sb[0]; R.sy0 = fg.sb[1]; R.sw = sh.w; R.sh = sh.h;
    R.seq_x0 = fg.sb[0] + fg.pad_x; R.seq_y0 = fg.sb[1] + fg.pad_y;
    R.seq_w = (uint32_t)(s->desc.xres + 2 * fg.pad_x); R.seq_h = (uint32_t)(s->desc.yres + 2 * fg.pad_y);
    R.max_lum = r->max_sample_luminance > 0.f ? r->max_sample_luminance : std::numeric_limits<float>::infinity();
    R.filter_rx = fg.rx; R.filter_ry = fg.ry;
    R.acc = fg.wide ? (unsigned long long *)d_slab : nullptr;
    const bool sobol_nd = r->sampler == PBRT_HIP_SAMPLER_SOBOL_ND || r->sampler == PBRT_HIP_SAMPLER_HALTON;  // the table samplers' instantiation
    RenderScratch rs;
    rc = ensure_render_scratch(s, r, fg, sh, &rs);  // (no allocation when pbrt_hip_render_prepare ran for this description, or an earlier frame did)
    if (rc) return rc;
    R.sobol_mat = r->sampler == PBRT_HIP_SAMPLER_HALTON ? s->d_halton.p : (sobol_nd ? s->d_sobol.p : nullptr);
    R.tri_uv = s->d_tri_uv.p;
    R.textures = s->d_textures.p;
    R.integrator = r->integrator;
    R.max_depth = r->max_depth;
    R.spp_x = r->spp_x;
    R.spp_y = r->spp_y;
    R.seed = r->seed;
    R.rank = r->rank;
    R.world = r->world_size;
    R.inv_nx = 1.0f / (float)r->spp_x;
    R.inv_ny = 1.0f / (float)r->spp_y;
    R.counters = s->d_counters.p;
    R.sampler = r->sampler;
    const uint32_t spp = r->spp_x * r->spp_y;
    R.spp_mask = 0;
    while (R.spp_mask + 1u < spp) R.spp_mask = 2u * R.spp_mask + 1u;
    // reciprocals for the kernel's two divisions by run-time values (kernels.hip pixel_xy, stratified sample): ceil(2^32 / d);
    // tsup / stx is exact while tsup * stx < 2^32
    auto recip32 = [](uint32_t d) { return d <= 1u ? 0u : (uint32_t)(((1ull << 32) + d - 1u) / d); };
    R.stx_recip = recip32(sh.stx);
    R.spp_x_recip = recip32(r->spp_x);
    if ((uint64_t)sh.total * (uint64_t)sh.stx >= (1ull << 32))
      return fail(PBRT_HIP_ERR_LIMIT, "render: film too large for the kernel's tile arithmetic");
    // The render kernel's waves are persistent: as many one-wave workgroups as the device holds at once (20 per CU: the
    // LDS stack -- render_stack_plan -- and the register budget both allow 5 per SIMD), each lane drawing item after item
    // from the rank's list.
    // An item is one CHUNK (a K-th of the samples, K <= 16 with at least 32 samples per chunk) of one pixel: DESIGN.md 3.1.
    R.chunk_shift = rs.chunk_shift;
    const uint32_t n_chunks = 1u << R.chunk_shift;  // K: DESIGN.md 3.1
    R.n_workgroups = rs.n_workgroups;
    R.next_item = reinterpret_cast<uint32_t *>(s->d_counters.p + 8);  // 8 counters, 64 bytes apart
    R.n_regions = std::min<uint32_t>(8u, std::max<uint32_t>(1u, tuning("PBRT_HIP_REGIONS", 8u, 8)));
    R.lane_state = s->d_lane_state.p;
    R.wide_slots = s->d_lane_state.p + (size_t)R.n_workgroups * 320;
    R.partials = fg.wide ? nullptr : s->d_partials.p;
    R.stack_overflow = s->d_stack_overflow.p;
    R.stack_overflow_entries = rs.plan.extra_entries;
    R.min_walkers = tuning("PBRT_HIP_MIN_WALKERS", s->dev.quad_stack_need <= kShallowStackNeed ? kMinWalkersShallow : kMinWalkers);
    R.min_parked = tuning("PBRT_HIP_MIN_PARKED", kMinParked);
    const int counters = (r->flags & PBRT_HIP_FLAG_COUNTERS) ? 1 : ((r->flags & PBRT_HIP_FLAG_WALK_COUNTERS) ? 2 : 0);
    if (counters == 1) {  // the canonical walk: a device-built scene gets the oracle's tree now (host build, first use only)
      const int ce = ensure_canonical(s);
      if (ce) return ce;
    }
    HIP_TRY(hipMemsetAsync(s->d_counters.p, 0, 80 * sizeof(unsigned long long), st));
    if (fg.wide && fg.crop_px()) HIP_TRY(hipMemsetAsync(d_slab, 0, fg.crop_px() * 32, st));  // this rank's accumulators start at zero
    HIP_TRY(hipEventRecord(s->ev0, st));
    // one launch per pass (one pass unless the partial sums would pass the cap: partials_passes) renders every item of the pass; the merge
    // adds each pixel's K partial sums in chunk order (a wide filter has no partial sums: its samples go straight into the accumulators)
    for (uint32_t pass = 0; pass < rs.passes; pass++) {
      const uint32_t n_pass = sh.n_local > pass ? (sh.n_local - pass + rs.passes - 1) / rs.passes : 0;
      if (n_pass == 0 && pass > 0) break;
      R.rank = r->rank + r->world_size * pass;
      R.world = r->world_size * rs.passes;
      R.n_items = n_pass * 4096u * n_chunks;
      if (pass > 0)  // the hand-out positions start again; the ray counters (the first 64 bytes) run on
        HIP_TRY(hipMemsetAsync(s->d_counters.p + 8, 0, 72 * sizeof(unsigned long long), st));
      HIP_TRY(launch_render(counters == 1 ? s->dev_exact : s->dev, R, n_pass, s->bvh.depth, counters, fg.wide, sobol_nd, st, r->integrator == PBRT_HIP_INTEGRATOR_PATH_MIS, s->textured));
      if (!fg.wide) HIP_TRY(launch_merge(R.partials, (float4 *)d_slab, sh.w, sh.h, R.rank, R.world, n_pass, spp, st, pass, rs.passes));
    }
    HIP_TRY(hipEventRecord(s->ev1, st));
    s->pending = true;
    s->pending_counters = counters != 0;
    // samples = pixels of this rank's super-tiles that lie inside the film
    uint64_t px = 0;
    for (uint32_t j = 0; j < sh.n_local; j++) {
      uint32_t t = r->rank + j * r->world_size;
      int32_t x0 = (int32_t)(t % sh.stx) * 64, y0 = (int32_t)(t / sh.stx) * 64;
      int32_t w = sh.w - x0 < 64 ? sh.w - x0 : 64, h = sh.h - y0 < 64 ? sh.h - y0 : 64;
      px += (uint64_t)w * (uint64_t)h;
    }
    s->pending_samples = px * (uint64_t)r->spp_x * (uint64_t)r->spp_y;
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_render_wait(pbrt_hip_scene *s, pbrt_hip_stats *stats) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "render_wait: null scene");
  if (!s->pending) return fail(PBRT_HIP_ERR_INVALID, "render_wait: no render in flight");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->ev1));
  s->pending = false;
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->samples = s->pending_samples;
    if (s->pending_counters) {
      unsigned long long c[5];
      HIP_TRY(hipMemcpy(c, s->d_counters.p, sizeof(c), hipMemcpyDeviceToHost));
      stats->camera_rays = c[0]; stats->bounce_rays = c[1]; stats->shadow_rays = c[2];
      stats->nodes_visited = c[3]; stats->tris_tested = c[4];
    }
  }
  return PBRT_HIP_OK;
}

int pbrt_hip_film_assemble_device(const pbrt_hip_scene *s, const void *d_slab, uint32_t rank, uint32_t world,
                                  void *d_film, void *stream) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "film_assemble: null argument");
  if (world == 0 || rank >= world) return fail(PBRT_HIP_ERR_INVALID, "film_assemble: rank must be < world_size");
  const Shard sh = make_shard(s->desc.xres, s->desc.yres, s->desc.crop, rank, world);
  if (sh.w <= 0 || sh.h <= 0) return PBRT_HIP_OK;  // (an empty crop window: a film of no pixels, which a caller may well hold in a NULL buffer)
  if (!d_film) return fail(PBRT_HIP_ERR_INVALID, "film_assemble: null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (sh.n_local && !d_slab) return fail(PBRT_HIP_ERR_INVALID, "film_assemble: null slab");
  HIP_TRY(launch_assemble((const float4 *)d_slab, (float4 *)d_film, sh.w, sh.h, rank, world, sh.n_local,
                          (hipStream_t)stream));
  return PBRT_HIP_OK;
}

int pbrt_hip_render(pbrt_hip_scene *s, const pbrt_hip_render_desc *r, float *film, pbrt_hip_stats *stats) {
  int rc = check_render_desc(s, r);
  if (rc) return rc;
  if (!film) return fail(PBRT_HIP_ERR_INVALID, "render: null film");
  HIP_TRY(hipSetDevice(s->device));
  const FilmGeom fg = film_geom(s->desc, *r);
  const Shard sh = make_shard_bounds(fg.sb, r->rank, r->world_size);
  const size_t n_px = fg.crop_px();
  const size_t slab_n = fg.wide ? 2 * n_px : (size_t)sh.n_local * 4096;  // (wide: four int64 accumulators per pixel = two float4)
  if (s->d_slab.n < slab_n) { s->d_slab.release(); HIP_TRY(s->d_slab.alloc(slab_n)); }
  if (s->d_film.n < n_px) { s->d_film.release(); HIP_TRY(s->d_film.alloc(n_px)); }
  if (n_px) HIP_TRY(hipMemsetAsync(s->d_film.p, 0, n_px * 16, s->stream));
  rc = pbrt_hip_render_device(s, r, s->d_slab.p, s->stream);
  if (rc) return rc;
  if (!n_px) rc = PBRT_HIP_OK;  // (an empty crop window: nothing was sampled, there is no film to assemble -- an empty film like the oracle's, not an error)
  else if (fg.wide) rc = pbrt_hip_film_from_acc_device(s, s->d_slab.p, s->d_film.p, s->stream);
  else rc = pbrt_hip_film_assemble_device(s, s->d_slab.p, r->rank, r->world_size, s->d_film.p, s->stream);
  hipError_t e = hipSuccess;
  if (!rc && n_px) e = hipMemcpyAsync(film, s->d_film.p, n_px * 16, hipMemcpyDeviceToHost, s->stream);
  const hipError_t e2 = hipStreamSynchronize(s->stream);  // (also on failure: the scene must not stay "in flight")
  if (rc || e != hipSuccess || e2 != hipSuccess) {
    s->pending = false;
    if (rc) return rc;
    return fail(PBRT_HIP_ERR_HIP, std::string("render: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  }
  return pbrt_hip_render_wait(s, stats);
}

void pbrt_hip_sobol_matrices(uint32_t *out) { sobol_nd_matrices(out); }

int64_t pbrt_hip_render_buffer_bytes(const pbrt_hip_scene *s, const pbrt_hip_render_desc *r) {
  if (!s || !r || r->world_size == 0 || r->rank >= r->world_size) return -1;
  const FilmGeom fg = film_geom(s->desc, *r);
  if (fg.wide) return (int64_t)fg.crop_px() * 32;
  return (int64_t)make_shard_bounds(fg.sb, r->rank, r->world_size).n_local * 4096 * 16;
}

int pbrt_hip_film_from_acc_device(const pbrt_hip_scene *s, const void *d_acc, void *d_film, void *stream) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "film_from_acc: null argument");
  int32_t b[4];
  film_cropped_bounds(s->desc.xres, s->desc.yres, s->desc.crop, b);
  const size_t n_px = (size_t)std::max(0, b[2] - b[0]) * (size_t)std::max(0, b[3] - b[1]);
  if (!n_px) return PBRT_HIP_OK;  // (an empty crop window)
  if (!d_film) return fail(PBRT_HIP_ERR_INVALID, "film_from_acc: null argument");
  if (n_px && !d_acc) return fail(PBRT_HIP_ERR_INVALID, "film_from_acc: null accumulators");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(launch_film_from_acc((const unsigned long long *)d_acc, (float4 *)d_film, n_px, (hipStream_t)stream));
  return PBRT_HIP_OK;
}

int pbrt_hip_render_acc(pbrt_hip_scene *s, const pbrt_hip_render_desc *r, int64_t *acc, pbrt_hip_stats *stats) {
  int rc = check_render_desc(s, r);
  if (rc) return rc;
  const FilmGeom fg = film_geom(s->desc, *r);
  if (!fg.wide) return fail(PBRT_HIP_ERR_INVALID, "render_acc: the default box filter has no accumulators (use pbrt_hip_render)");
  const size_t n_px = fg.crop_px();
  if (n_px && !acc) return fail(PBRT_HIP_ERR_INVALID, "render_acc: null output");
  HIP_TRY(hipSetDevice(s->device));
  if (s->d_slab.n < 2 * n_px) { s->d_slab.release(); HIP_TRY(s->d_slab.alloc(2 * n_px)); }
  rc = pbrt_hip_render_device(s, r, s->d_slab.p, s->stream);
  if (rc) return rc;
  hipError_t e = n_px ? hipMemcpyAsync(acc, s->d_slab.p, n_px * 32, hipMemcpyDeviceToHost, s->stream) : hipSuccess;
  const hipError_t e2 = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || e2 != hipSuccess) {
    s->pending = false;
    return fail(PBRT_HIP_ERR_HIP, std::string("render_acc: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  }
  return pbrt_hip_render_wait(s, stats);
}

// host restatement of film_from_acc_kernel (kernels.hip), for hosts that add the accumulators of several ranks themselves
void pbrt_hip_film_from_acc(const int64_t *acc, int64_t n_px, float *film) {
  const float inv = 1.0f / kFixedOne;
  for (int64_t i = 0; i < n_px; i++) {
    const float r = (float)acc[4 * i] * inv, g = (float)acc[4 * i + 1] * inv, b = (float)acc[4 * i + 2] * inv;
    film[4 * i] = 0.412453f * r + 0.357580f * g + 0.180423f * b;
    film[4 * i + 1] = 0.212671f * r + 0.715160f * g + 0.072169f * b;
    film[4 * i + 2] = 0.019334f * r + 0.119193f * g + 0.950227f * b;
    film[4 * i + 3] = (float)acc[4 * i + 3];
  }
}

int64_t pbrt_hip_slab_floats(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world) {
  if (!crop || world == 0 || rank >= world || xres <= 0 || yres <= 0) return -1;
  return (int64_t)make_shard(xres, yres, crop, rank, world).n_local * 4096 * 4;
}

int pbrt_hip_slab_pixel_index(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world,
                              int64_t *out) {
  if (!crop || !out || world == 0 || rank >= world || xres <= 0 || yres <= 0)
    return fail(PBRT_HIP_ERR_INVALID, "slab_pixel_index: bad argument");
  const Shard sh = make_shard(xres, yres, crop, rank, world);
  for (uint32_t j = 0; j < sh.n_local; j++) {
    const uint32_t t = rank + j * world;
    const int32_t x0 = (int32_t)(t % sh.stx) * 64, y0 = (int32_t)(t / sh.stx) * 64;
    for (int32_t py = 0; py < 64; py++)
      for (int32_t px = 0; px < 64; px++) {
        const int32_t x = x0 + px, y = y0 + py;
        out[(size_t)j * 4096 + py * 64 + px] = (x < sh.w && y < sh.h) ? (int64_t)y * sh.w + x : -1;
      }
  }
  return PBRT_HIP_OK;
}

static int ray_batch(pbrt_hip_scene *s, int64_t n, const float *o, const float *d, const float *tmax, float *t,
                     uint32_t *prim, float *b1, float *b2, uint8_t *occ, uint64_t *counters, bool any) {
  if (!s) return fail(PBRT_HIP_ERR_INVALID, "intersect: null scene");
  if (n < 0 || (n && (!o || !d || !tmax))) return fail(PBRT_HIP_ERR_INVALID, "intersect: bad ray arrays");
  if (n == 0) {
    if (counters) counters[0] = counters[1] = 0;
    return PBRT_HIP_OK;
  }
  HIP_TRY(hipSetDevice(s->device));
  if (counters) {
    const int ce = ensure_canonical(s);
    if (ce) return ce;
  }
  DevBuf<float> d_o, d_d, d_tmax, d_t, d_b1, d_b2;
  DevBuf<uint32_t> d_prim;
  DevBuf<uint8_t> d_occ;
  int rc = PBRT_HIP_OK;
  auto cleanup = [&]() {
    d_o.release(); d_d.release(); d_tmax.release(); d_t.release(); d_b1.release(); d_b2.release();
    d_prim.release(); d_occ.release();
  };
#define RB_TRY(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      cleanup();                                                                                  \
      return fail(PBRT_HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    }                                                                                             \
  } while (0)
  RB_TRY(d_o.alloc(3 * (size_t)n));
  RB_TRY(d_d.alloc(3 * (size_t)n));
  RB_TRY(d_tmax.alloc((size_t)n));
  RB_TRY(hipMemcpyAsync(d_o.p, o, 12 * (size_t)n, hipMemcpyHostToDevice, s->stream));
  RB_TRY(hipMemcpyAsync(d_d.p, d, 12 * (size_t)n, hipMemcpyHostToDevice, s->stream));
  RB_TRY(hipMemcpyAsync(d_tmax.p, tmax, 4 * (size_t)n, hipMemcpyHostToDevice, s->stream));
  RayBatch B{};
  B.o = d_o.p; B.d = d_d.p; B.tmax = d_tmax.p; B.n = n;
  B.min_walkers = tuning("PBRT_HIP_MIN_WALKERS", kMinWalkers);
  B.min_parked = tuning("PBRT_HIP_MIN_PARKED", kMinParked);
  if (any) {
    RB_TRY(d_occ.alloc((size_t)n));
    B.occluded = d_occ.p;
  } else {
    RB_TRY(d_t.alloc((size_t)n)); RB_TRY(d_prim.alloc((size_t)n)); RB_TRY(d_b1.alloc((size_t)n)); RB_TRY(d_b2.alloc((size_t)n));
    B.t = d_t.p; B.prim = d_prim.p; B.b1 = d_b1.p; B.b2 = d_b2.p;
  }
  if (counters) {
    RB_TRY(hipMemsetAsync(s->d_counters.p, 0, 2 * sizeof(unsigned long long), s->stream));
    B.counters = s->d_counters.p;
  }
  {
    // launch_intersect uses at most 4096 workgroups of 4 waves
    const uint32_t extra = s->dev.quad_stack_need + 2u > kIntersectLdsStack ? s->dev.quad_stack_need + 2u - kIntersectLdsStack : 0;  // sentinel + entries beyond the kIntersectLdsStack - 1 kept in LDS
    const size_t need = (size_t)4096 * 4 * 64 * extra;
    if (s->d_stack_overflow.n < need) { s->d_stack_overflow.release(); RB_TRY(s->d_stack_overflow.alloc(need)); }
    B.stack_overflow = s->d_stack_overflow.p;
    B.stack_overflow_entries = extra;
  }
  const bool timed = debug_knob("PBRT_HIP_TIME_INTERSECT") != nullptr;  // tuning aid: kernel time to stderr
  if (timed) RB_TRY(hipEventRecord(s->ev0, s->stream));
  RB_TRY(launch_intersect(counters ? s->dev_exact : s->dev, B, any, s->bvh.depth, s->stream));
  if (timed) {
    RB_TRY(hipEventRecord(s->ev1, s->stream));
    RB_TRY(hipEventSynchronize(s->ev1));
    float ms = 0.f;
    RB_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    std::fprintf(stderr, "pbrt_hip intersect kernel: %lld rays %.3f ms %.1f Mrays/s\n", (long long)n, ms, (double)n / ms / 1e3);
  }
  if (any) {
    RB_TRY(hipMemcpyAsync(occ, d_occ.p, (size_t)n, hipMemcpyDeviceToHost, s->stream));
  } else {
    RB_TRY(hipMemcpyAsync(t, d_t.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    RB_TRY(hipMemcpyAsync(prim, d_prim.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    RB_TRY(hipMemcpyAsync(b1, d_b1.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    RB_TRY(hipMemcpyAsync(b2, d_b2.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s->stream));
  }
  if (counters) {
    unsigned long long c[2];
    RB_TRY(hipMemcpyAsync(c, s->d_counters.p, sizeof(c), hipMemcpyDeviceToHost, s->stream));
    RB_TRY(hipStreamSynchronize(s->stream));
    counters[0] = c[0];
    counters[1] = c[1];
  } else {
    RB_TRY(hipStreamSynchronize(s->stream));
  }
#undef RB_TRY
  cleanup();
  return rc;
}

int pbrt_hip_intersect(pbrt_hip_scene *s, int64_t n, const float *o, const float *d, const float *tmax, float *t,
                       uint32_t *prim, float *b1, float *b2, uint64_t *counters) {
  if (n > 0 && (!t || !prim || !b1 || !b2)) return fail(PBRT_HIP_ERR_INVALID, "intersect: null output array");
  return ray_batch(s, n, o, d, tmax, t, prim, b1, b2, nullptr, counters, false);
}

int pbrt_hip_occluded(pbrt_hip_scene *s, int64_t n, const float *o, const float *d, const float *tmax, uint8_t *hit) {
  if (n > 0 && !hit) return fail(PBRT_HIP_ERR_INVALID, "occluded: null output array");
  return ray_batch(s, n, o, d, tmax, nullptr, nullptr, nullptr, nullptr, hit, nullptr, true);
}

// ---- host pieces ----
void pbrt_hip_film_cropped_bounds(int32_t xres, int32_t yres, const float crop[4], int32_t out[4]) {
  film_cropped_bounds(xres, yres, crop, out);
}

// Film::get_sample_bounds, core/film.rs:166-175
void pbrt_hip_film_sample_bounds(int32_t xres, int32_t yres, const float crop[4], float rx, float ry, int32_t out[4]) {
  int32_t c[4];
  film_cropped_bounds(xres, yres, crop, c);
  out[0] = (int32_t)std::floor((float)c[0] + 0.5f - rx);
  out[1] = (int32_t)std::floor((float)c[1] + 0.5f - ry);
  out[2] = (int32_t)std::ceil((float)c[2] - 0.5f + rx);
  out[3] = (int32_t)std::ceil((float)c[3] - 0.5f + ry);
}

// Film::get_film_tile, core/film.rs:264-281
void pbrt_hip_film_tile_bounds(int32_t xres, int32_t yres, const float crop[4], float rx, float ry, const int32_t sb[4],
                               int32_t out[4]) {
  int32_t c[4];
  film_cropped_bounds(xres, yres, crop, c);
  const int32_t x0 = (int32_t)std::ceil((float)sb[0] - 0.5f - rx), y0 = (int32_t)std::ceil((float)sb[1] - 0.5f - ry);
  const int32_t x1 = (int32_t)(std::floor((float)sb[2] - 0.5f + rx) + 1.f);
  const int32_t y1 = (int32_t)(std::floor((float)sb[3] - 0.5f + ry) + 1.f);
  out[0] = x0 > c[0] ? x0 : c[0];
  out[1] = y0 > c[1] ? y0 : c[1];
  out[2] = x1 < c[2] ? x1 : c[2];
  out[3] = y1 < c[3] ? y1 : c[3];
}

// Film::write_image's pixel loop, core/film.rs:346-372 (splat_xyz is never written: add_splat is
// unimplemented!() at film.rs:334-336, so the splat term is identically zero)
void pbrt_hip_film_to_rgb(const float *film, int64_t n, float scale, float *rgb) {
  for (int64_t i = 0; i < n; i++) {
    float c[3];
    xyz_to_rgb(film + 4 * i, c);
    const float w = film[4 * i + 3];
    if (w != 0.f) {
      const float inv = 1.f / w;
      for (int k = 0; k < 3; k++) {
        const float v = c[k] * inv;
        c[k] = v > 0.f ? v : 0.f;
      }
    }
    for (int k = 0; k < 3; k++) rgb[3 * i + k] = c[k] * scale;
  }
}

void pbrt_hip_look_at(const float pos[3], const float look[3], const float up[3], float m[16], float m_inv[16]) {
  look_at(pos, look, up, m, m_inv);
}

}  // extern "C"

// ---- scene ingestion (scene_parser.cpp) ----
struct pbrt_hip_loaded {
  pbrt_hip::LoadedScene s;
};

namespace {
const char *parse_error_name(pbrt_hip::ParseError e) {
  switch (e) {
    case pbrt_hip::ParseError::Eof: return "Eof";
    case pbrt_hip::ParseError::UnterminatedString: return "UnterminatedString";
    case pbrt_hip::ParseError::MixedParameters: return "MixedParameters";
    case pbrt_hip::ParseError::Unquoted: return "Unquoted";
    case pbrt_hip::ParseError::Syntax: return "Syntax";
    case pbrt_hip::ParseError::NotImplemented: return "NotImplemented";
    case pbrt_hip::ParseError::Io: return "Io";
    default: return "None";
  }
}
// "<kind>: <the reference's Display text for that kind, parser.rs:31-58>[: detail]"
std::string parse_error_text(pbrt_hip::ParseError e, const std::string &msg) {
  const std::string kind = parse_error_name(e);
  switch (e) {
    case pbrt_hip::ParseError::Eof: return kind + ": premature EOF" + (msg.empty() ? "" : " (" + msg + ")");
    case pbrt_hip::ParseError::UnterminatedString: return kind + ": unterminated string";
    case pbrt_hip::ParseError::MixedParameters: return kind + ": mixed string and numeric parameters";
    case pbrt_hip::ParseError::Unquoted: return kind + ": expected quoted string" + (msg.empty() ? "" : " (" + msg + ")");
    case pbrt_hip::ParseError::Syntax:
      return kind + (msg.rfind("input not float", 0) == 0 ? ": " + msg : ": syntax error: '" + msg + "'");
    case pbrt_hip::ParseError::NotImplemented: return kind + ": have not yet implemented '" + msg + "'";
    default: return kind + ": " + msg;
  }
}
size_t copy_out(const std::string &s, char *buf, size_t cap) {
  if (buf && cap) {
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return s.size();
}
}  // namespace

extern "C" {

int pbrt_hip_load_string(const char *text, size_t len, const char *base_dir, pbrt_hip_loaded **out) {
  if (!text || !out) return fail(PBRT_HIP_ERR_INVALID, "load_string: null argument");
  *out = nullptr;
  try {
    std::unique_ptr<pbrt_hip_loaded> l(new pbrt_hip_loaded());
    std::string msg;
    pbrt_hip::ParseError e = pbrt_hip::parse_scene(text, len, base_dir ? base_dir : "", &l->s, &msg);
    if (e != pbrt_hip::ParseError::None) return fail(PBRT_HIP_ERR_INVALID, parse_error_text(e, msg));
    *out = l.release();
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_load_file(const char *path, pbrt_hip_loaded **out) {
  if (!path || !out) return fail(PBRT_HIP_ERR_INVALID, "load_file: null argument");
  *out = nullptr;
  FILE *f = std::fopen(path, "rb");
  if (!f) return fail(PBRT_HIP_ERR_INVALID, std::string("Io: cannot open '") + path + "'");  // api.rs:392-395
  try {
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    f = nullptr;
    std::string p(path);
    size_t slash = p.rfind('/');
    std::string dir = slash == std::string::npos ? "" : p.substr(0, slash);
    return pbrt_hip_load_string(text.data(), text.size(), dir.c_str(), out);
  } catch (const std::exception &e) {
    if (f) std::fclose(f);
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

void pbrt_hip_loaded_free(pbrt_hip_loaded *l) { delete l; }

int pbrt_hip_loaded_get(const pbrt_hip_loaded *l, pbrt_hip_scene_desc *d, pbrt_hip_render_desc *r, char *filename,
                        size_t cap) {
  if (!l) return fail(PBRT_HIP_ERR_INVALID, "loaded_get: null scene");
  const pbrt_hip::LoadedScene &s = l->s;
  if (d) {
    std::memset(d, 0, sizeof *d);
    d->P = s.P.data(); d->idx = s.idx.data(); d->mat_id = s.mat_id.data();
    d->mats = s.mats.data(); d->lights = s.lights.data(); d->spheres = s.spheres.data();
    d->n_verts = (uint32_t)(s.P.size() / 3); d->n_tris = (uint32_t)s.mat_id.size(); d->n_mats = (uint32_t)s.mats.size();
    d->n_lights = (uint32_t)s.lights.size(); d->n_spheres = (uint32_t)s.spheres.size();
    std::memcpy(d->cam_to_world, s.cam_to_world, 64);
    d->fov = s.fov; d->xres = s.xres; d->yres = s.yres;
    std::memcpy(d->crop, s.crop, 16);
    d->tri_uv = s.tri_uv.empty() ? nullptr : s.tri_uv.data();
    d->textures = s.textures.empty() ? nullptr : s.textures.data();
    d->n_textures = (uint32_t)s.textures.size();
  }
  if (r) {
    std::memset(r, 0, sizeof *r);
    r->integrator = s.integrator; r->max_depth = s.max_depth; r->spp_x = s.spp_x; r->spp_y = s.spp_y;
    r->seed = 0; r->rank = 0; r->world_size = 1;
    r->sampler = s.sampler;
    r->filter_xwidth = s.filter_radius[0]; r->filter_ywidth = s.filter_radius[1];
    r->max_sample_luminance = s.max_sample_luminance;
  }
  copy_out(s.filename, filename, cap);
  return PBRT_HIP_OK;
}

float pbrt_hip_loaded_film_scale(const pbrt_hip_loaded *l) { return l ? l->s.film_scale : 1.f; }

int pbrt_hip_loaded_warnings(const pbrt_hip_loaded *l, char *buf, size_t cap) {
  if (!l) return 0;
  std::string all;
  for (const std::string &w : l->s.warnings) { all += w; all += '\n'; }
  copy_out(all, buf, cap);
  return (int)l->s.warnings.size();
}

int pbrt_hip_loaded_state(const pbrt_hip_loaded *l, float ctm[16], char *names, size_t cap) {
  if (!l) return fail(PBRT_HIP_ERR_INVALID, "loaded_state: null scene");
  const pbrt_hip::LoadedScene &s = l->s;
  if (ctm) std::memcpy(ctm, s.final_ctm, 64);
  copy_out(s.camera_name + " " + s.sampler_name + " " + s.integrator_name + " " + s.filter_name + " " +
               s.accelerator_name + " " + s.film_name, names, cap);
  return PBRT_HIP_OK;
}

int pbrt_hip_tokenize(const char *text, size_t len, char *buf, size_t cap) {
  if (!text) return -1;
  pbrt_hip::Tokenizer t(text, len);
  std::string tok, all;
  int n = 0;
  pbrt_hip::ParseError e;
  while (t.next(&tok, &e)) {
    if (e != pbrt_hip::ParseError::None) { copy_out(all, buf, cap); return -(1 + n); }
    all += tok;
    all += '\n';
    n++;
  }
  copy_out(all, buf, cap);
  return n;
}

}  // extern "C"
