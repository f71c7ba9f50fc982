// sbvh_build.cpp -- see sbvh_build.hpp.  One stack of references; a node owns the top `n` of it, partitions them in
// place (duplicates of split references are appended), and its two children consume them: no pointer tree, nodes are
// emitted in pre-order (first child = next node) like bvh_build.cpp.
#include "sbvh_build.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace pbrt_hip {
namespace {

constexpr float kInfF = std::numeric_limits<float>::infinity();
constexpr uint32_t kMedianBelowLevel = 56;  // deeper than this: halve by position only (bounds the depth)

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; a++) { lo[a] = kInfF; hi[a] = -kInfF; }
  }
  void grow(const float *l, const float *h) {
    for (int a = 0; a < 3; a++) {
      if (l[a] < lo[a]) lo[a] = l[a];
      if (h[a] > hi[a]) hi[a] = h[a];
    }
  }
  void grow(const Box &b) { grow(b.lo, b.hi); }
  bool empty() const { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }
  float area() const {  // half the surface area; 0 for an empty box
    if (empty()) return 0.f;
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx * dy + dx * dz) + dy * dz;
  }
  void clip_to(const Box &b) {
    for (int a = 0; a < 3; a++) {
      if (lo[a] < b.lo[a]) lo[a] = b.lo[a];
      if (hi[a] > b.hi[a]) hi[a] = b.hi[a];
    }
  }
};

struct Ref {
  uint32_t tri;
  Box b;
};

// largest float <= x and smallest float >= x, one more ulp outwards each (the double arithmetic that produced x has an
// error far below a float ulp, the extra step covers it)
inline float round_down(double x) {
  float f = (float)x;
  if ((double)f > x) f = std::nextafterf(f, -kInfF);
  return std::nextafterf(f, -kInfF);
}
inline float round_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafterf(f, kInfF);
  return std::nextafterf(f, kInfF);
}

struct Builder {
  const float *P;
  const uint32_t *idx;
  SbvhParams prm;
  RefBvh *out;
  std::vector<Ref> refs;  // the stack
  float root_area = 0.f;
  // scratch of the sweeps
  std::vector<float> right_area;
  std::vector<Box> right_box;

  const float *vert(uint32_t tri, int k) const { return P + 3 * (size_t)idx[3 * (size_t)tri + k]; }

  // The two references a plane (axis a, position pos) makes of r: the boxes of the parts of the TRIANGLE on either
  // side, clipped to r's own box.  Intersections of edges with the plane in double, rounded outwards.
  void split_ref(const Ref &r, int a, float pos, Ref *l, Ref *rr) const {
    Box L, R;
    L.reset();
    R.reset();
    for (int e = 0; e < 3; e++) {
      const float *v0 = vert(r.tri, e), *v1 = vert(r.tri, (e + 1) % 3);
      const float p0 = v0[a], p1 = v1[a];
      if (p0 <= pos) L.grow(v0, v0);
      if (p0 >= pos) R.grow(v0, v0);
      if ((p0 < pos && p1 > pos) || (p0 > pos && p1 < pos)) {
        const double t = ((double)pos - (double)p0) / ((double)p1 - (double)p0);
        float plo[3], phi[3];
        for (int k = 0; k < 3; k++) {
          const double x = (double)v0[k] + t * ((double)v1[k] - (double)v0[k]);
          plo[k] = round_down(x);
          phi[k] = round_up(x);
        }
        plo[a] = phi[a] = pos;
        L.grow(plo, phi);
        R.grow(plo, phi);
      }
    }
    // widen by a fraction of the triangle's own extent: a ray that Moeller-Trumbore's rounding lets hit a hair outside
    // an edge must still pass through the box that holds that edge (the whole-triangle box has the same exposure at its
    // faces); then clip to the box of the reference that is being split, which is as wide or wider there
    float tlo[3], thi[3];
    for (int k = 0; k < 3; k++) {
      const float x0 = vert(r.tri, 0)[k], x1 = vert(r.tri, 1)[k], x2 = vert(r.tri, 2)[k];
      tlo[k] = std::min(x0, std::min(x1, x2));
      thi[k] = std::max(x0, std::max(x1, x2));
    }
    for (int k = 0; k < 3; k++) {
      const float w = (thi[k] - tlo[k]) * prm.pad;
      L.lo[k] -= w; L.hi[k] += w;
      R.lo[k] -= w; R.hi[k] += w;
    }
    L.hi[a] = pos;
    R.lo[a] = pos;
    L.clip_to(r.b);
    R.clip_to(r.b);
    l->tri = rr->tri = r.tri;
    l->b = L;
    rr->b = R;
  }

  struct ObjectSplit {
    float cost = kInfF;
    int axis = -1;
    uint32_t n_left = 0;   // sweep: references left of the plane in sorted order
    int bin = -1;          // binned: last bin of the left side
    float c0 = 0.f, scale = 0.f;
    Box lb, rb;
  };
  struct SpatialSplit {
    float cost = kInfF;
    int axis = -1;
    float pos = 0.f;
  };

  static float key_of(const Ref &r, int a) { return r.b.lo[a] + r.b.hi[a]; }
  static int bin_of(float key, float c0, float scale, int nb) {
    const float f = (key - c0) * scale;
    return f >= (float)(nb - 1) ? nb - 1 : (f > 0.f ? (int)f : 0);  // (a NaN goes to bin 0)
  }

  void sort_axis(Ref *r, uint32_t n, int a) const {
    std::sort(r, r + n, [a](const Ref &x, const Ref &y) {
      const float kx = key_of(x, a), ky = key_of(y, a);
      if (kx != ky) return kx < ky;
      if (x.tri != y.tri) return x.tri < y.tri;
      return x.b.lo[a] < y.b.lo[a];
    });
  }

  int widest_axis(const Ref *r, uint32_t n) const {
    float cmin[3] = {kInfF, kInfF, kInfF}, cmax[3] = {-kInfF, -kInfF, -kInfF};
    for (uint32_t i = 0; i < n; i++)
      for (int a = 0; a < 3; a++) {
        const float c = key_of(r[i], a);
        if (c < cmin[a]) cmin[a] = c;
        if (c > cmax[a]) cmax[a] = c;
      }
    const float ex = cmax[0] - cmin[0], ey = cmax[1] - cmin[1], ez = cmax[2] - cmin[2];
    return (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
  }

  ObjectSplit find_object_sweep(Ref *r, uint32_t n) {
    ObjectSplit best;
    if (right_area.size() < n) { right_area.resize(n); right_box.resize(n); }
    const int only = prm.widest_axis_only ? widest_axis(r, n) : -1;
    for (int a = 0; a < 3; a++) {
      if (only >= 0 && a != only) continue;
      sort_axis(r, n, a);
      Box acc;
      acc.reset();
      for (uint32_t i = n - 1; i > 0; i--) {
        acc.grow(r[i].b);
        right_area[i - 1] = acc.area();
        right_box[i - 1] = acc;
      }
      acc.reset();
      for (uint32_t i = 1; i < n; i++) {
        acc.grow(r[i - 1].b);
        const float c = acc.area() * (float)i + right_area[i - 1] * (float)(n - i);
        if (c < best.cost) { best.cost = c; best.axis = a; best.n_left = i; best.lb = acc; best.rb = right_box[i - 1]; }
      }
    }
    return best;
  }

  ObjectSplit find_object_binned(const Ref *r, uint32_t n) {
    ObjectSplit best;
    const int NB = prm.object_bins;
    float cmin[3] = {kInfF, kInfF, kInfF}, cmax[3] = {-kInfF, -kInfF, -kInfF};
    for (uint32_t i = 0; i < n; i++)
      for (int a = 0; a < 3; a++) {
        const float c = key_of(r[i], a);
        if (c < cmin[a]) cmin[a] = c;
        if (c > cmax[a]) cmax[a] = c;
      }
    std::vector<Box> bb((size_t)NB), rbx((size_t)NB);
    std::vector<uint32_t> cnt((size_t)NB);
    const int only = prm.widest_axis_only ? widest_axis(r, n) : -1;
    for (int a = 0; a < 3; a++) {
      if (only >= 0 && a != only) continue;
      if (!(cmax[a] > cmin[a])) continue;
      const float scale = (float)NB / (cmax[a] - cmin[a]);
      if (!std::isfinite(scale)) continue;  // (a denormal extent: no plane to speak of)
      for (int b = 0; b < NB; b++) { bb[b].reset(); cnt[b] = 0; }
      for (uint32_t i = 0; i < n; i++) {
        int b = bin_of(key_of(r[i], a), cmin[a], scale, NB);
        bb[b].grow(r[i].b);
        cnt[b]++;
      }
      Box acc;
      acc.reset();
      for (int b = NB - 1; b > 0; b--) { acc.grow(bb[b]); rbx[b - 1] = acc; }
      acc.reset();
      uint32_t nl = 0;
      for (int b = 0; b < NB - 1; b++) {
        acc.grow(bb[b]);
        nl += cnt[b];
        if (nl == 0 || nl == n) continue;
        const float c = acc.area() * (float)nl + rbx[b].area() * (float)(n - nl);
        if (c < best.cost) { best.cost = c; best.axis = a; best.bin = b; best.n_left = nl; best.c0 = cmin[a]; best.scale = scale; best.lb = acc; best.rb = rbx[b]; }
      }
    }
    return best;
  }

  SpatialSplit find_spatial(const Ref *r, uint32_t n, const Box &box) {
    SpatialSplit best;
    const int NB = prm.spatial_bins;
    std::vector<Box> bb((size_t)NB), rbx((size_t)NB);
    std::vector<uint32_t> enter((size_t)NB), leave((size_t)NB);
    for (int a = 0; a < 3; a++) {
      const float origin = box.lo[a], ext = box.hi[a] - box.lo[a];
      if (!(ext > 0.f)) continue;
      const float bin_size = ext / (float)NB, inv = 1.0f / bin_size;
      if (!(bin_size > 0.f) || !std::isfinite(inv)) continue;
      for (int b = 0; b < NB; b++) { bb[b].reset(); enter[b] = leave[b] = 0; }
      for (uint32_t i = 0; i < n; i++) {
        const int first = bin_of(r[i].b.lo[a], origin, inv, NB), last = std::max(first, bin_of(r[i].b.hi[a], origin, inv, NB));
        Ref cur = r[i];
        for (int b = first; b < last; b++) {
          Ref l, rr;
          split_ref(cur, a, origin + bin_size * (float)(b + 1), &l, &rr);
          bb[b].grow(l.b);
          cur = rr;
        }
        bb[last].grow(cur.b);
        enter[first]++;
        leave[last]++;
      }
      Box acc;
      acc.reset();
      for (int b = NB - 1; b > 0; b--) { acc.grow(bb[b]); rbx[b - 1] = acc; }
      acc.reset();
      uint32_t nl = 0, nr = n;
      for (int b = 1; b < NB; b++) {
        acc.grow(bb[b - 1]);
        nl += enter[b - 1];
        nr -= leave[b - 1];
        if (nl == 0 || nr == 0) continue;
        const float c = acc.area() * (float)nl + rbx[b - 1].area() * (float)nr;
        if (c < best.cost) { best.cost = c; best.axis = a; best.pos = origin + bin_size * (float)b; }
      }
    }
    return best;
  }

  void emit_leaf(uint32_t me, const Ref &r) {
    BvhNode nd;
    for (int a = 0; a < 3; a++) { nd.lo[a] = r.b.lo[a]; nd.hi[a] = r.b.hi[a]; }
    nd.offset = (uint32_t)out->ref_tri.size();
    nd.count_axis = 1u;
    out->ref_tri.push_back(r.tri);
    for (int a = 0; a < 3; a++) { out->ref_lo.push_back(r.b.lo[a]); out->ref_hi.push_back(r.b.hi[a]); }
    out->nodes[me] = nd;
  }

  // the top n references of the stack; `budget` = duplicates this subtree may still make
  uint32_t build(uint32_t n, uint32_t level, uint32_t budget) {
    const uint32_t me = (uint32_t)out->nodes.size();
    out->nodes.emplace_back();
    if (level + 1 > out->depth) out->depth = level + 1;
    if (n == 1) {
      emit_leaf(me, refs.back());
      refs.pop_back();
      return me;
    }
    const size_t base = refs.size() - n;
    Box box;
    box.reset();
    for (uint32_t i = 0; i < n; i++) box.grow(refs[base + i].b);

    uint32_t n_left = 0, n_right = 0;  // after the partition: [base, base + n_left) and the n_right above it
    bool done = false;
    int axis = 0;
    if (level < kMedianBelowLevel) {
      const bool sweep = n <= prm.sweep_below;
      ObjectSplit os = sweep ? find_object_sweep(&refs[base], n) : find_object_binned(&refs[base], n);
      SpatialSplit ss;
      if (budget > 0 && os.axis >= 0) {
        Box ov = os.lb;
        ov.clip_to(os.rb);
        if (ov.area() > prm.alpha * root_area) ss = find_spatial(&refs[base], n, box);
      } else if (budget > 0) {
        ss = find_spatial(&refs[base], n, box);
      }
      if (ss.axis >= 0 && ss.cost * prm.spatial_bias < os.cost) {
        // ---- spatial split: entirely-left | straddling | entirely-right, then every straddler goes left, goes right,
        // or is split, whichever makes the children cheapest (reference unsplitting) ----
        const int a = ss.axis;
        const float pos = ss.pos;
        size_t left_end = base, right_start = base + n, end = base + n;
        for (size_t i = base; i < right_start; i++) {
          if (refs[i].b.hi[a] <= pos) std::swap(refs[i], refs[left_end++]);
          else if (refs[i].b.lo[a] >= pos) std::swap(refs[i--], refs[--right_start]);
        }
        Box lb, rb;
        lb.reset();
        rb.reset();
        for (size_t i = base; i < left_end; i++) lb.grow(refs[i].b);
        for (size_t i = right_start; i < end; i++) rb.grow(refs[i].b);
        uint32_t made = 0;
        while (left_end < right_start) {
          const Ref cur = refs[left_end];
          Ref l, r;
          split_ref(cur, a, pos, &l, &r);
          Box lub = lb, rub = rb, ldb = lb, rdb = rb;
          lub.grow(l.b);
          rub.grow(r.b);
          ldb.grow(cur.b);
          rdb.grow(cur.b);
          const float nl = (float)(left_end - base), nr = (float)(end - right_start);
          const float lac = lb.area() * nl, rac = rb.area() * nr;
          const float to_left = ldb.area() * (nl + 1.f) + rac, to_right = lac + rdb.area() * (nr + 1.f);
          const float dup = made < budget ? lub.area() * (nl + 1.f) + rub.area() * (nr + 1.f) : kInfF;
          if (dup < to_left && dup < to_right && !l.b.empty() && !r.b.empty()) {
            lb = lub;
            rb = rub;
            refs[left_end++] = l;
            refs.push_back(r);
            end++;
            made++;
          } else if (to_left <= to_right) {
            lb = ldb;
            left_end++;
          } else {
            rb = rdb;
            std::swap(refs[left_end], refs[--right_start]);
          }
        }
        n_left = (uint32_t)(left_end - base);
        n_right = (uint32_t)(end - right_start);
        if (n_left > 0 && n_right > 0) {
          done = true;
          axis = a;
          budget -= made;
        }
        // (a side came out empty: the references are still all there -- split ones as two halves -- and the object
        // split below partitions them)
        if (!done) {
          n = (uint32_t)(refs.size() - base);
          budget -= std::min(budget, made);
          os = n <= prm.sweep_below ? find_object_sweep(&refs[base], n) : find_object_binned(&refs[base], n);
        }
      }
      if (!done && os.axis >= 0) {
        axis = os.axis;
        if (n <= prm.sweep_below) {
          sort_axis(&refs[base], n, os.axis);
          n_left = os.n_left;
        } else {
          const float c0 = os.c0, scale = os.scale;
          const int NB = prm.object_bins, bin = os.bin, ax = os.axis;
          auto it = std::partition(refs.begin() + base, refs.begin() + base + n, [&](const Ref &r) { return bin_of(key_of(r, ax), c0, scale, NB) <= bin; });
          n_left = (uint32_t)(it - (refs.begin() + base));
        }
        n_right = n - n_left;
        done = n_left > 0 && n_right > 0;
      }
    }
    if (!done) {  // no plane separates anything (coincident references) or very deep: halve as it stands
      n = (uint32_t)(refs.size() - base);
      n_left = n / 2;
      n_right = n - n_left;
    }
    // The top of the stack is processed first and becomes child 0 (the next node).  The walk pops the children it stacked
    // in slot order, so which side comes first matters (with this scene's rays -- camera +y, light above -- the low side
    // first measures 15 % fewer node steps than the high side first): the low part is rotated to the top.
    const uint32_t total = n_left + n_right;
    const uint32_t b_right = (uint32_t)((uint64_t)budget * n_right / total), b_left = budget - b_right;
    if (prm.low_side_first) {
      std::rotate(refs.begin() + base, refs.begin() + base + n_left, refs.end());
      build(n_left, level + 1, b_left);
      const uint32_t second = build(n_right, level + 1, b_right);
      BvhNode nd;
      for (int a = 0; a < 3; a++) { nd.lo[a] = box.lo[a]; nd.hi[a] = box.hi[a]; }
      nd.offset = second;
      nd.count_axis = (uint32_t)axis << 16;
      out->nodes[me] = nd;
      return me;
    }
    build(n_right, level + 1, b_right);
    const uint32_t second = build(n_left, level + 1, b_left);
    BvhNode nd;
    for (int a = 0; a < 3; a++) { nd.lo[a] = box.lo[a]; nd.hi[a] = box.hi[a]; }
    nd.offset = second;
    nd.count_axis = (uint32_t)axis << 16;
    out->nodes[me] = nd;
    return me;
  }
};

}  // namespace

void build_sbvh(const float *P, const uint32_t *idx, uint32_t n_tris, const SbvhParams &prm, RefBvh *out) {
  out->nodes.clear();
  out->ref_tri.clear();
  out->ref_lo.clear();
  out->ref_hi.clear();
  out->depth = 0;
  if (n_tris == 0) return;
  Builder b;
  b.P = P;
  b.idx = idx;
  b.prm = prm;
  b.out = out;
  b.refs.reserve((size_t)((double)n_tris * (1.0 + prm.budget)) + 64);
  Box root;
  root.reset();
  for (uint32_t t = 0; t < n_tris; t++) {
    Ref r;
    r.tri = t;
    r.b.reset();
    for (int k = 0; k < 3; k++) r.b.grow(b.vert(t, k), b.vert(t, k));
    root.grow(r.b);
    b.refs.push_back(r);
  }
  b.root_area = root.area();
  out->nodes.reserve(2 * b.refs.capacity());
  b.build(n_tris, 0, (uint32_t)((double)n_tris * prm.budget));
}

void refs_of_bvh(const Bvh &bv, const float *P, const uint32_t *idx, RefBvh *out) {
  out->nodes = bv.nodes;
  out->depth = bv.depth;
  const size_t n = bv.order.size();
  out->ref_tri = bv.order;
  out->ref_lo.resize(3 * n);
  out->ref_hi.resize(3 * n);
  for (size_t r = 0; r < n; r++) {
    const uint32_t t = bv.order[r];
    for (int a = 0; a < 3; a++) {
      const float v0 = P[3 * (size_t)idx[3 * (size_t)t] + a], v1 = P[3 * (size_t)idx[3 * (size_t)t + 1] + a], v2 = P[3 * (size_t)idx[3 * (size_t)t + 2] + a];
      out->ref_lo[3 * r + a] = std::min(v0, std::min(v1, v2));
      out->ref_hi[3 * r + a] = std::max(v0, std::max(v1, v2));
    }
  }
}


// ---- global optimisation of a built tree by re-insertion (sbvh_build.hpp): an interior node is taken out (its sibling moves up), and
// its two children are put back where they add the least surface area, found by branch and bound.  Leaves (one reference) stay whole.
namespace {
struct ON { float lo[3], hi[3]; int parent, l, r; uint32_t ref; };
inline float on_area(const float *lo, const float *hi) { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return (dx * dy + dx * dz) + dy * dz; }
inline float on_union_area(const ON &a, const ON &b) {
  float lo[3], hi[3];
  for (int k = 0; k < 3; k++) { lo[k] = std::min(a.lo[k], b.lo[k]); hi[k] = std::max(a.hi[k], b.hi[k]); }
  return on_area(lo, hi);
}
}  // namespace

void reinsert_optimize(RefBvh *t, int passes, float frac) {
  const int n = (int)t->nodes.size();
  if (n < 7) return;
  std::vector<ON> T((size_t)n);
  for (int i = 0; i < n; i++) {
    const BvhNode &b = t->nodes[i];
    ON &o = T[i];
    for (int k = 0; k < 3; k++) { o.lo[k] = b.lo[k]; o.hi[k] = b.hi[k]; }
    o.parent = -1;
    if (b.count_axis & 0xffffu) { o.l = o.r = -1; o.ref = b.offset; }
    else { o.l = i + 1; o.r = (int)b.offset; o.ref = 0; }
  }
  for (int i = 0; i < n; i++) if (T[i].l >= 0) { T[T[i].l].parent = i; T[T[i].r].parent = i; }
  int root = 0;
  auto refit_up = [&](int i) {
    for (; i >= 0; i = T[i].parent) {
      const ON &a = T[T[i].l], &b = T[T[i].r];
      for (int k = 0; k < 3; k++) { T[i].lo[k] = std::min(a.lo[k], b.lo[k]); T[i].hi[k] = std::max(a.hi[k], b.hi[k]); }
    }
  };
  auto replace_child = [&](int p, int from, int to) {
    if (p < 0) { root = to; T[to].parent = -1; return; }
    if (T[p].l == from) T[p].l = to; else T[p].r = to;
    T[to].parent = p;
  };
  struct QE { float c; int node; bool operator<(const QE &o) const { return c > o.c; } };
  std::vector<QE> heap;
  auto find_best = [&](int x) {
    const float ax = on_area(T[x].lo, T[x].hi);
    float best_cost = std::numeric_limits<float>::infinity();
    int best = root;
    heap.clear();
    heap.push_back({0.f, root});
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end());
      const QE e = heap.back();
      heap.pop_back();
      if (e.c + ax >= best_cost) break;
      const ON &y = T[e.node];
      const float direct = on_union_area(y, T[x]), total = e.c + direct;
      if (total < best_cost) { best_cost = total; best = e.node; }
      const float down = total - on_area(y.lo, y.hi);
      if (y.l >= 0 && down + ax < best_cost) {
        heap.push_back({down, y.l}); std::push_heap(heap.begin(), heap.end());
        heap.push_back({down, y.r}); std::push_heap(heap.begin(), heap.end());
      }
    }
    return best;
  };
  std::vector<int> cand;
  for (int pass = 0; pass < passes; pass++) {
    cand.clear();
    for (int i = 0; i < n; i++)
      if (T[i].l >= 0 && i != root && T[i].parent != root) cand.push_back(i);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return on_area(T[a].lo, T[a].hi) > on_area(T[b].lo, T[b].hi); });
    const size_t take = std::max<size_t>(1, (size_t)(frac * (float)cand.size()));
    for (size_t ci = 0; ci < take && ci < cand.size(); ci++) {
      const int N = cand[ci];
      const int P = T[N].parent;
      if (N == root || P < 0 || P == root || T[N].l < 0) continue;  // (the tree has changed under the list)
      const int G = T[P].parent, S = T[P].l == N ? T[P].r : T[P].l;
      int A = T[N].l, B = T[N].r;
      if (on_area(T[A].lo, T[A].hi) < on_area(T[B].lo, T[B].hi)) std::swap(A, B);
      replace_child(G, P, S);
      refit_up(G);
      const int freeN[2] = {N, P}, sub[2] = {A, B};
      for (int k = 0; k < 2; k++) {
        const int X = sub[k], F = freeN[k], best = find_best(X), bp = T[best].parent;
        replace_child(bp, best, F);
        T[F].l = best; T[F].r = X;
        T[best].parent = F; T[X].parent = F;
        refit_up(F);
      }
    }
  }
  // flatten (depth first, child 0 = the child whose centre is lower along the axis that separates the two centres most)
  RefBvh out;
  out.nodes.reserve((size_t)n);
  out.ref_tri.reserve(t->ref_tri.size());
  struct It { int node; int patch; uint32_t level; };
  std::vector<It> st = {{root, -1, 1u}};
  while (!st.empty()) {
    const It it = st.back();
    st.pop_back();
    const int me = (int)out.nodes.size();
    if (it.patch >= 0) out.nodes[it.patch].offset = (uint32_t)me;
    if (it.level > out.depth) out.depth = it.level;
    const ON &o = T[it.node];
    BvhNode b;
    for (int k = 0; k < 3; k++) { b.lo[k] = o.lo[k]; b.hi[k] = o.hi[k]; }
    if (o.l < 0) {
      b.offset = (uint32_t)out.ref_tri.size();
      b.count_axis = 1u;
      out.ref_tri.push_back(t->ref_tri[o.ref]);
      for (int k = 0; k < 3; k++) { out.ref_lo.push_back(t->ref_lo[3 * (size_t)o.ref + k]); }
      for (int k = 0; k < 3; k++) { out.ref_hi.push_back(t->ref_hi[3 * (size_t)o.ref + k]); }
      out.nodes.push_back(b);
    } else {
      int c0 = o.l, c1 = o.r, ax = 0;
      float sep = -1.f;
      for (int k = 0; k < 3; k++) {
        const float d = std::fabs((T[c0].lo[k] + T[c0].hi[k]) - (T[c1].lo[k] + T[c1].hi[k]));
        if (d > sep) { sep = d; ax = k; }
      }
      if ((T[c0].lo[ax] + T[c0].hi[ax]) > (T[c1].lo[ax] + T[c1].hi[ax])) std::swap(c0, c1);
      b.offset = 0;
      b.count_axis = (uint32_t)ax << 16;
      out.nodes.push_back(b);
      st.push_back({c1, me, it.level + 1});  // right: patched when reached
      st.push_back({c0, -1, it.level + 1});  // left: the next node
    }
  }
  *t = std::move(out);
}

}  // namespace pbrt_hip
