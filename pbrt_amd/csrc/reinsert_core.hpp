// reinsert_core.hpp -- one pass of PARALLEL RE-INSERTION over a built binary BVH (the global optimisation of the production
// walk's tree), written once for the device (bvh_gpu.hip: one thread per node, one launch per phase)
// and for the host (reinsert_batch.cpp: the same functions called in a loop -- what the CPU tests and tools/walk_sim.py
// run, so that the logic is exercised where no GPU exists).  After Meister & Bittner, "Parallel Reinsertion for Bounding
// Volume Hierarchy Optimization" (Eurographics 2018); the sequential form (Bittner, Hapala, Havran 2013) was round 3's
// opt-in host pass.  The reference has no accelerator at all (core/api.rs:237: the name "bvh" is stored; the render call is
// a comment, api.rs:446-453), so there is nothing to conform to but the RESULT: by the tie rule of DESIGN.md 3.4 a hit does
// not depend on the tree.
//
// A pass has five phases, each a parallel loop over nodes with a barrier (a kernel boundary) after it:
//   1. SEARCH   every node x looks, in the tree as it stands (read-only), for the position that lowers the summed surface
//               area of the interior nodes most if x were taken out (its parent p disappears, its sibling moves up, the
//               ancestors shrink) and put back as the sibling of some node y under the freed p: find_move().  The search
//               starts where x is and climbs: at every ancestor ("pivot") it descends into the subtree on the other side
//               with branch and bound -- the area saved so far minus the area the nodes between pivot and y would grow by
//               minus area(x u y) -- stackless (parent links), with a cap on the nodes visited, so that coincident boxes,
//               where nothing prunes, cost a bounded amount (ADVICE / DESIGN r03: the host pass's known limit).
//   2. LOCK     a move rewrites the links of x, its parent, sibling and grandparent and of y and its parent: its key (gain, x)
//               goes into each of the six with atomicMax.
//   3. CHECK    a move whose key survived in ALL six holds them; of those, a move with the node of another holder OF LARGER KEY
//               on the way from its target up to the common ancestor is dropped (target_path_is_free: no cycle can form).  The others are
//               applied; dropped moves search again next pass.  No link is written twice, and the outcome does not depend on
//               the order the threads run in: deterministic.  (Locking the whole paths, as Meister & Bittner do, would also
//               make the gains of a pass's moves add up exactly -- but in a soup of triangles as large as their spacing one
//               node in eight wants to move ACROSS the top of the tree, and every such move's path runs through the root's
//               children: measured, 3 650 of 25 052 moves of the first pass survive and the pass converges to a worse tree.)
//   4. APPLY    seven link writes per move.
//   5. REFIT    the boxes above the moved nodes, bottom-up (the device: only the chains above a move's old and new place -- a pass
//               moves about one node in seventy, bvh_gpu.hip ri_mark_kernel / ri_refit_dirty_kernel; the host run: all of them;
//               the same boxes either way, min / max being exact).  Then the summed area of the interior nodes in fixed point: a
//               pass that RAISED it is undone and ends the passes (the moves of one pass lock six nodes each, not their
//               paths, so their gains need not add up: one pass in a thousand or so comes out worse).
// Node ids: interior nodes [0, n_int), the leaf of slot k = n_int + k (n_int = leaves - 1), root 0.  The root and its two
// children are never moved (the root stays node 0); they can be targets.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RI_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RI_HD inline
#endif

namespace pbrt_hip {
namespace reins {

constexpr uint32_t kNone = 0xffffffffu;

struct Box {
  float lo[3], hi[3];
};

// the tree being optimised (all arrays in the memory of whoever runs the pass)
struct Tree {
  uint32_t n_int;           // interior nodes; n_int + 1 leaves
  uint32_t *par;            // [2 n_int + 1] parent (root: kNone)
  uint32_t *kid;            // [2 n_int] the two children of interior node i: kid[2 i], kid[2 i + 1]
  unsigned long long *bx;   // [3 (2 n_int + 1)] a box as {lo.x lo.y}{lo.z hi.x}{hi.y hi.z} (bvh_gpu.hip's layout)
};

RI_HD float bits_f(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
RI_HD uint32_t f_bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}
RI_HD Box load_box(const Tree &t, uint32_t i) {
  const unsigned long long w0 = t.bx[3 * (size_t)i], w1 = t.bx[3 * (size_t)i + 1], w2 = t.bx[3 * (size_t)i + 2];
  Box b;
  b.lo[0] = bits_f((uint32_t)w0); b.lo[1] = bits_f((uint32_t)(w0 >> 32));
  b.lo[2] = bits_f((uint32_t)w1); b.hi[0] = bits_f((uint32_t)(w1 >> 32));
  b.hi[1] = bits_f((uint32_t)w2); b.hi[2] = bits_f((uint32_t)(w2 >> 32));
  return b;
}
RI_HD void store_box(const Tree &t, uint32_t i, const Box &b) {
  t.bx[3 * (size_t)i] = (unsigned long long)f_bits(b.lo[0]) | ((unsigned long long)f_bits(b.lo[1]) << 32);
  t.bx[3 * (size_t)i + 1] = (unsigned long long)f_bits(b.lo[2]) | ((unsigned long long)f_bits(b.hi[0]) << 32);
  t.bx[3 * (size_t)i + 2] = (unsigned long long)f_bits(b.hi[1]) | ((unsigned long long)f_bits(b.hi[2]) << 32);
}
RI_HD float fmin_(float a, float b) { return a < b ? a : b; }
RI_HD float fmax_(float a, float b) { return a > b ? a : b; }
RI_HD Box unite(const Box &a, const Box &b) {
  Box u;
  for (int k = 0; k < 3; k++) { u.lo[k] = fmin_(a.lo[k], b.lo[k]); u.hi[k] = fmax_(a.hi[k], b.hi[k]); }
  return u;
}
RI_HD float area(const Box &b) {  // half the surface area
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return (dx * dy + dx * dz) + dy * dz;
}
RI_HD uint32_t sibling(const Tree &t, uint32_t parent, uint32_t child) {
  const uint32_t a = t.kid[2 * (size_t)parent], b = t.kid[2 * (size_t)parent + 1];
  return a == child ? b : a;
}

// When the passes end: ONE rule for the device loop (bvh_gpu.hip) and the host run (reinsert_batch.cpp), so that both make the same
// tree from the same input whatever the number of passes it takes.
struct StopRule {
  int max_passes = 12;
  uint32_t min_moved_div = 1024;            // a pass that applied fewer than nodes / 1024 moves was the last one
  unsigned long long visit_budget = 1024;   // ... as is one after which the searches have looked at more than this many nodes PER NODE of the
                                            // tree in total (coincident boxes, where nothing prunes a search, cost a bounded time)
  uint32_t min_tris = 1024;                 // trees of fewer triangles are left as built (nothing measurable to gain)
};
inline bool stop_after_pass(const StopRule &r, unsigned long long applied, unsigned long long visits_total, uint32_t n_nodes) {
  return applied * r.min_moved_div < n_nodes || visits_total > r.visit_budget * n_nodes;
}

struct Move {
  uint32_t y;       // x becomes the sibling of y (kNone: x stays)
  uint32_t lca;     // the node neither of whose links or box the move changes: the paths x -> lca and y -> lca are what it locks
  float gain;       // summed interior surface area the move removes (> 0)
  uint32_t visits;  // nodes the search looked at (statistics)
};

// Phase 1.
// What a search is steered by.
struct Search {
  uint32_t max_visits = 512;  // cap on the nodes one search looks at (typical: 35 on average, 200 at most, for 1M random triangles)
  float min_rel = 1e-4f;      // a move must gain more than this x area(parent of x): float noise, and moves not worth a conflict
  // The moved node ends up as a child slot of a quad node whose 8-BIT GRID spans about its new parent's box, so the walk sees
  // x about one cell wider per axis: qk x the new parent's extent (2 / 255).  The cost of a move counts the surface area x
  // gains that way, weighted qw for a triangle (the collapse's c_tri) and 1 for an interior node.  Without the term a soup of
  // triangles smaller than a top-level cell (1M triangles: 0.010 against 0.016) hangs triangles that straddle a split high
  // in the tree: 11 % MORE triangle tests per ray than the unoptimised tree instead of 3 % fewer (tools/experiments/README.md).
  float qk = 2.0f / 255.0f, qw = 2.0f;
};

// the surface area of box b as a grid of cell size k x extent(q) per axis holds it
RI_HD float area_on(const Box &b, const Box &q, float k) {
  const float dx = (b.hi[0] - b.lo[0]) + k * (q.hi[0] - q.lo[0]), dy = (b.hi[1] - b.lo[1]) + k * (q.hi[1] - q.lo[1]), dz = (b.hi[2] - b.lo[2]) + k * (q.hi[2] - q.lo[2]);
  return (dx * dy + dx * dz) + dy * dz;
}
RI_HD Move find_move(const Tree &t, uint32_t x, const Search &sp) {
  const uint32_t max_visits = sp.max_visits;
  const float min_rel = sp.min_rel, qk = sp.qk, qw = sp.qw;
  Move m;
  m.y = kNone; m.lca = kNone; m.gain = 0.f; m.visits = 0;
  const uint32_t p = t.par[x];
  if (p == kNone || t.par[p] == kNone) return m;  // the root and its children stay
  const Box bxx = load_box(t, x);
  const float ax = area(bxx);
  // The growth `c` of the nodes above the candidate is kept in FIXED POINT (units of 2^-se, the root's area below 2^40 units): the
  // stackless descent adds a node's term on the way down and takes the same term off on the way up, and integer addition undoes
  // exactly -- a float sum drifted by an ulp per level over the hundreds of nodes a search may visit, and with it the bound and the
  // gain that orders the locks.  A tree whose root area is not finite (coordinates near 1e30) has nothing to steer by.
  const float a_root = area(load_box(t, 0u));
  if (!(a_root < 3.0e38f)) return m;
  int se = 0;
  (void)frexpf(a_root, &se);
  se = 40 - se;
  se = se > 100 ? 100 : (se < -100 ? -100 : se);
  const float to_fix = ldexpf(1.0f, se), from_fix = ldexpf(1.0f, -se);
  const Box bp0 = load_box(t, p);
  float saved = area(bp0);  // taking x out: p disappears (below: + what the ancestors under the pivot shrink by)
  float best = min_rel * saved;
  const float wx = x < t.n_int ? 1.f : qw;
  const float q_old = qk > 0.f ? wx * area_on(bxx, bp0, qk) : 0.f, q_min = qk > 0.f ? wx * area_on(bxx, bxx, qk) : 0.f;
  uint32_t pivot = p, other = sibling(t, p, x), visits = 0;
  Box nb;  // the box of the pivot's x-side child once x is gone
  nb.lo[0] = nb.lo[1] = nb.lo[2] = nb.hi[0] = nb.hi[1] = nb.hi[2] = 0.f;
  bool first = true;
  for (;;) {
    // branch and bound in the subtree of `other`: c = what the nodes from `other` down to the parent of `out` grow by
    Box top = load_box(t, other);
    {
      uint32_t out = other;
      long long ci = 0;
      bool down = true;
      for (;;) {
        if (down) {
          const Box bo = out == other ? top : load_box(t, out);
          visits++;
          const Box un = unite(bo, bxx);
          const float direct = area(un);
          const float c = (float)ci * from_fix;
          if (!(first && out == other)) {  // (x's own sibling: putting x back where it was)
            float g = (saved - c) - direct;
            if (qk > 0.f) g -= wx * area_on(bxx, un, qk) - q_old;
            if (g > best) { best = g; m.y = out; m.lca = pivot; }
          }
          const long long di = (long long)((direct - area(bo)) * to_fix);
          const float cn = (float)(ci + di) * from_fix;
          if (out < t.n_int && ((saved - cn) - ax) - (q_min - q_old) > best && visits < max_visits) {
            ci += di;
            out = t.kid[2 * (size_t)out];
            continue;
          }
          down = false;
        }
        if (out == other) break;
        const uint32_t q = t.par[out];
        if (out == t.kid[2 * (size_t)q]) {
          out = t.kid[2 * (size_t)q + 1];
          down = true;
        } else {
          const Box bq = load_box(t, q);
          ci -= (long long)((area(unite(bq, bxx)) - area(bq)) * to_fix);  // the term q added on the way down, bit for bit
          out = q;
        }
      }
    }
    if (visits >= max_visits) break;
    nb = first ? top : unite(nb, top);  // the pivot's box without x
    const uint32_t up = t.par[pivot];
    if (up == kNone) break;
    if (!first) {
      // x as the sibling of the pivot itself: the pivot shrinks to nb, the freed node above it gets the pivot's old box
      const float an = area(nb);
      float g = saved - an;
      if (qk > 0.f) g -= wx * area_on(bxx, load_box(t, pivot), qk) - q_old;
      if (g > best) { best = g; m.y = pivot; m.lca = up; }
      saved += area(load_box(t, pivot)) - an;
    }
    other = sibling(t, up, pivot);
    pivot = up;
    first = false;
  }
  m.gain = m.y == kNone ? 0.f : best;
  m.visits = visits;
  return m;
}

RI_HD unsigned long long move_key(uint32_t x, float gain) { return ((unsigned long long)f_bits(gain) << 32) | x; }  // gain > 0: its bits order like it

// the nodes whose LINKS a move writes (apply_move): f(node) for each
template <class F>
RI_HD void for_move_nodes(const Tree &t, uint32_t x, uint32_t y, F &&f) {
  const uint32_t p = t.par[x];
  f(x);
  f(p);
  f(sibling(t, p, x));
  f(t.par[p]);
  f(y);
  f(t.par[y]);
}

// Moves whose link sets are disjoint can still form a CYCLE together (x1 into the subtree of x2 and x2 into the subtree of
// x1).  A cycle needs a move whose target y has, strictly between it and the common ancestor with x, another moving node:
// such a move is dropped -- when that other move has the LARGER key.  (Proof sketch: in the tree after the moves, a node's chain
// of ancestors climbs the old tree except where it passes a moved node x' and jumps to the parent of its target; a closed chain
// must enter some moved x' from below, i.e. through a target inside its old subtree, and not all the moved nodes of the chain
// can be nested in each other: the moves of a closed chain block one another in a ring.  In every such ring the move in front of
// the one with the largest key is blocked by a larger key and dropped, so no ring survives whole.  Until round 5 a move was dropped
// whatever the blocker's key: both moves of a mutually crossing pair went, and the same pair could be found and dropped pass
// after pass -- ADVICE r04.)
// blocks(q): q is the node of another move that holds its link locks AND whose key is larger than this move's.
template <class F>
RI_HD bool target_path_is_free(const Tree &t, uint32_t x, uint32_t y, uint32_t lca, F &&blocks) {
  for (uint32_t q = t.par[y]; q != lca && q != kNone; q = t.par[q])
    if (q != x && blocks(q)) return false;
  return true;
}

// Phase 4 (only for moves that hold all their locks).  Children of the re-used node p: (y, x).
RI_HD void apply_move(const Tree &t, uint32_t x, uint32_t y) {
  const uint32_t p = t.par[x], s = sibling(t, p, x), g = t.par[p], yp = t.par[y];
  const size_t gs = 2 * (size_t)g + (t.kid[2 * (size_t)g] == p ? 0 : 1), ys = 2 * (size_t)yp + (t.kid[2 * (size_t)yp] == y ? 0 : 1);
  t.kid[gs] = s;
  t.par[s] = g;
  t.kid[ys] = p;
  t.par[p] = yp;
  t.kid[2 * (size_t)p] = y;
  t.kid[2 * (size_t)p + 1] = x;
  t.par[y] = p;
}

// After the last pass: child 0 of every interior node = the child whose centre is lower along the axis that separates the
// two centres most (the walk enters the nearer hit child and stacks the others in slot order; sbvh_build.hpp).
RI_HD void order_children(const Tree &t, uint32_t i) {
  const uint32_t c0 = t.kid[2 * (size_t)i], c1 = t.kid[2 * (size_t)i + 1];
  const Box a = load_box(t, c0), b = load_box(t, c1);
  int ax = 0;
  float sep = -1.f;
  for (int k = 0; k < 3; k++) {
    const float d = (a.lo[k] + a.hi[k]) - (b.lo[k] + b.hi[k]);
    const float ad = d < 0.f ? -d : d;
    if (ad > sep) { sep = ad; ax = k; }
  }
  if ((a.lo[ax] + a.hi[ax]) > (b.lo[ax] + b.hi[ax])) {
    t.kid[2 * (size_t)i] = c1;
    t.kid[2 * (size_t)i + 1] = c0;
  }
}

}  // namespace reins
}  // namespace pbrt_hip
