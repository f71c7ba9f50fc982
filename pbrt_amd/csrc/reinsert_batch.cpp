// reinsert_batch.cpp -- the HOST run of parallel re-insertion (reinsert_core.hpp): the phases the device runs as kernels
// (bvh_gpu.hip), here as plain loops over the same functions.  It exists so that the logic of the device pass is exercised,
// and its trees are walked and costed (tests/test_host.py, tools/walk_sim.py, tools/fuzz), where no GPU exists; a lock is a
// 64-bit maximum whichever order the nodes are visited in, so the host's sequential loops and the device's threads produce
// the same tree from the same input.  Nothing here is on the product's render path: the product optimises on the device.
#include "reinsert_batch.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>


namespace pbrt_hip {

using reins::kNone;

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

void single_ref_tree(const Bvh &bv, const float *P, const uint32_t *idx, RefBvh *out) {
  RefBvh flat;
  refs_of_bvh(bv, P, idx, &flat);
  RefBvh o;
  o.ref_tri = flat.ref_tri;
  o.ref_lo = flat.ref_lo;
  o.ref_hi = flat.ref_hi;
  o.nodes.reserve(2 * flat.ref_tri.size());
  // (iterative: the canonical tree can be 64 levels deep, a leaf adds at most 6)
  struct It { uint32_t node, first, count; int patch; uint32_t level; };  // node: canonical node, or 0xffffffff for a run of references
  std::vector<It> st;
  if (!flat.nodes.empty()) st.push_back({0u, 0u, 0u, -1, 1u});
  while (!st.empty()) {
    const It it = st.back();
    st.pop_back();
    const int me = (int)o.nodes.size();
    if (it.patch >= 0) o.nodes[it.patch].offset = (uint32_t)me;
    if (it.level > o.depth) o.depth = it.level;
    BvhNode b{};
    uint32_t first = it.first, count = it.count;
    if (it.node != 0xffffffffu) {
      const BvhNode &c = flat.nodes[it.node];
      b = c;
      if ((c.count_axis & 0xffffu) == 0) {
        o.nodes.push_back(b);
        st.push_back({c.offset, 0u, 0u, me, it.level + 1});
        st.push_back({it.node + 1u, 0u, 0u, -1, it.level + 1});
        continue;
      }
      first = c.offset;
      count = c.count_axis & 0xffffu;
    }
    if (it.node == 0xffffffffu || count == 1u) {  // the box of a run: its references' boxes
      for (int a = 0; a < 3; a++) { b.lo[a] = flat.ref_lo[3 * (size_t)first + a]; b.hi[a] = flat.ref_hi[3 * (size_t)first + a]; }
      for (uint32_t r = first + 1; r < first + count; r++)
        for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], flat.ref_lo[3 * (size_t)r + a]); b.hi[a] = std::max(b.hi[a], flat.ref_hi[3 * (size_t)r + a]); }
    }
    if (count == 1u) {
      b.offset = first;
      b.count_axis = 1u;
      o.nodes.push_back(b);
    } else {
      b.offset = 0;
      b.count_axis &= 0xffff0000u;
      o.nodes.push_back(b);
      const uint32_t half = count / 2;
      st.push_back({0xffffffffu, first + half, count - half, me, it.level + 1});
      st.push_back({0xffffffffu, first, half, -1, it.level + 1});
    }
  }
  *out = std::move(o);
}

bool LinkTree::valid(std::string *why) const {
  const uint32_t n_nodes = 2 * n_int + 1;
  auto bad = [&](const char *m) { if (why) *why = m; return false; };
  if (par.size() != n_nodes || kid.size() != 2 * (size_t)n_int || bx.size() != 3 * (size_t)n_nodes) return bad("array sizes");
  if (par[0] != kNone) return bad("the root has a parent");
  std::vector<uint8_t> seen(n_nodes, 0);
  std::vector<uint32_t> st = {0u};
  size_t reached = 0;
  reins::Tree t{n_int, const_cast<uint32_t *>(par.data()), const_cast<uint32_t *>(kid.data()), const_cast<unsigned long long *>(bx.data())};
  while (!st.empty()) {
    const uint32_t i = st.back();
    st.pop_back();
    if (i >= n_nodes) return bad("child out of range");
    if (seen[i]) return bad("a node is reached twice");
    seen[i] = 1;
    reached++;
    if (i < n_int) {
      const reins::Box b = reins::load_box(t, i);
      for (int k = 0; k < 2; k++) {
        const uint32_t c = kid[2 * (size_t)i + k];
        if (c >= n_nodes) return bad("child out of range");
        if (par[c] != i) return bad("a child's parent link does not point back");
        const reins::Box cb = reins::load_box(t, c);
        for (int a = 0; a < 3; a++)
          if (cb.lo[a] < b.lo[a] || cb.hi[a] > b.hi[a]) return bad("a child's box is not inside its parent's");
        st.push_back(c);
      }
    }
  }
  if (reached != n_nodes) return bad("not every node is reachable from the root");
  return true;
}

double LinkTree::cost() const {
  reins::Tree t{n_int, const_cast<uint32_t *>(par.data()), const_cast<uint32_t *>(kid.data()), const_cast<unsigned long long *>(bx.data())};
  double c = 0;
  for (uint32_t i = 0; i < n_int; i++) c += reins::area(reins::load_box(t, i));
  return c;
}

void link_tree_of(const RefBvh &rb, LinkTree *out) {
  const size_t nn = rb.nodes.size();
  const uint32_t n_leaves = (uint32_t)rb.ref_tri.size();
  out->n_int = n_leaves ? n_leaves - 1 : 0;
  const uint32_t n_int = out->n_int, n_nodes = 2 * n_int + 1;
  out->par.assign(n_nodes, kNone);
  out->kid.assign(2 * (size_t)n_int, kNone);
  out->bx.assign(3 * (size_t)n_nodes, 0ull);
  reins::Tree t{n_int, out->par.data(), out->kid.data(), out->bx.data()};
  std::vector<uint32_t> id(nn);
  uint32_t next = 0;
  for (size_t i = 0; i < nn; i++) id[i] = (rb.nodes[i].count_axis & 0xffffu) ? n_int + rb.nodes[i].offset : next++;
  for (size_t i = 0; i < nn; i++) {
    const BvhNode &b = rb.nodes[i];
    reins::Box bb;
    for (int a = 0; a < 3; a++) { bb.lo[a] = b.lo[a]; bb.hi[a] = b.hi[a]; }
    reins::store_box(t, id[i], bb);
    if ((b.count_axis & 0xffffu) == 0) {
      const uint32_t c0 = id[i + 1], c1 = id[b.offset];
      out->kid[2 * (size_t)id[i]] = c0;
      out->kid[2 * (size_t)id[i] + 1] = c1;
      out->par[c0] = id[i];
      out->par[c1] = id[i];
    }
  }
}

void ref_bvh_of(const LinkTree &lt, const RefBvh &refs, RefBvh *out) {
  RefBvh o;
  const uint32_t n_int = lt.n_int;
  reins::Tree t{n_int, const_cast<uint32_t *>(lt.par.data()), const_cast<uint32_t *>(lt.kid.data()), const_cast<unsigned long long *>(lt.bx.data())};
  o.nodes.reserve(2 * (size_t)n_int + 1);
  o.ref_tri.reserve(refs.ref_tri.size());
  o.ref_lo.reserve(refs.ref_lo.size());
  o.ref_hi.reserve(refs.ref_hi.size());
  struct It { uint32_t node; int patch; uint32_t level; };
  std::vector<It> st = {{0u, -1, 1u}};
  while (!st.empty()) {
    const It it = st.back();
    st.pop_back();
    const int me = (int)o.nodes.size();
    if (it.patch >= 0) o.nodes[it.patch].offset = (uint32_t)me;
    if (it.level > o.depth) o.depth = it.level;
    const reins::Box bb = reins::load_box(t, it.node);
    BvhNode b;
    for (int a = 0; a < 3; a++) { b.lo[a] = bb.lo[a]; b.hi[a] = bb.hi[a]; }
    if (it.node >= n_int) {
      const uint32_t r = it.node - n_int;
      b.offset = (uint32_t)o.ref_tri.size();
      b.count_axis = 1u;
      o.ref_tri.push_back(refs.ref_tri[r]);
      for (int a = 0; a < 3; a++) o.ref_lo.push_back(refs.ref_lo[3 * (size_t)r + a]);
      for (int a = 0; a < 3; a++) o.ref_hi.push_back(refs.ref_hi[3 * (size_t)r + a]);
      o.nodes.push_back(b);
    } else {
      b.offset = 0;
      b.count_axis = 0u;
      o.nodes.push_back(b);
      st.push_back({lt.kid[2 * (size_t)it.node + 1], me, it.level + 1});  // second child: patched when reached
      st.push_back({lt.kid[2 * (size_t)it.node], -1, it.level + 1});      // first child: the next node
    }
  }
  *out = std::move(o);
}

void refit_links(LinkTree *lt) {
  const uint32_t n_int = lt->n_int;
  reins::Tree t{n_int, lt->par.data(), lt->kid.data(), lt->bx.data()};
  std::vector<uint32_t> order, st = {0u};
  order.reserve(n_int);
  while (!st.empty()) {
    const uint32_t i = st.back();
    st.pop_back();
    if (i >= n_int) continue;
    order.push_back(i);
    st.push_back(lt->kid[2 * (size_t)i]);
    st.push_back(lt->kid[2 * (size_t)i + 1]);
  }
  for (size_t k = order.size(); k-- > 0;) {
    const uint32_t i = order[k];
    reins::store_box(t, i, reins::unite(reins::load_box(t, lt->kid[2 * (size_t)i]), reins::load_box(t, lt->kid[2 * (size_t)i + 1])));
  }
}

void reinsert_batch_links(LinkTree *lt, const ReinsertBatchParams &prm, ReinsertBatchStats *stats) {
  const uint32_t n_int = lt->n_int, n_nodes = 2 * n_int + 1;
  if (stats) *stats = ReinsertBatchStats{};
  if (n_int < 3) return;
  reins::Tree t{n_int, lt->par.data(), lt->kid.data(), lt->bx.data()};
  std::vector<reins::Move> mv(n_nodes);
  std::vector<unsigned long long> lock(n_nodes);
  std::vector<uint8_t> holds(n_nodes);
  const bool verbose = std::getenv("PBRT_HIP_REINSERT_VERBOSE") != nullptr;
  const uint32_t mu = std::max(1u, prm.mu);
  if (const char *sq = std::getenv("PBRT_HIP_REINSERT_SEQ")) {  // experiment: the same moves applied one at a time
    auto refit_up = [&](uint32_t i) {
      for (; i != kNone; i = lt->par[i]) reins::store_box(t, i, reins::unite(reins::load_box(t, lt->kid[2 * (size_t)i]), reins::load_box(t, lt->kid[2 * (size_t)i + 1])));
    };
    std::vector<uint32_t> ord(n_nodes);
    for (int pass = 0; pass < prm.passes; pass++) {
      for (uint32_t i = 0; i < n_nodes; i++) ord[i] = i;
      if (sq[0] == 'a') std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return reins::area(reins::load_box(t, a)) > reins::area(reins::load_box(t, b)); });
      uint64_t applied = 0;
      for (uint32_t x : ord) {
        const reins::Move m = reins::find_move(t, x, prm.search);
        if (m.y == kNone) continue;
        const uint32_t g = lt->par[lt->par[x]];
        reins::apply_move(t, x, m.y);
        refit_up(g);
        refit_up(lt->par[x]);
        applied++;
      }
      if (verbose) std::fprintf(stderr, "sequential pass %d: %llu applied, cost %.6g\n", pass, (unsigned long long)applied, lt->cost());
    }
    for (uint32_t i = 0; i < n_int; i++) reins::order_children(t, i);
    return;
  }
  // the summed area of the interior nodes as the device reduces it: exact integers (units of 2^-se of reinsert_core.hpp's find_move)
  auto fixed_cost = [&]() {
    int se = 0;
    (void)frexpf(reins::area(reins::load_box(t, 0u)), &se);
    se = 40 - se;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    const float to_fix = ldexpf(1.0f, se);
    unsigned long long sum = 0;
    for (uint32_t i = 0; i < n_int; i++) sum += (unsigned long long)(reins::area(reins::load_box(t, i)) * to_fix);
    return sum;
  };
  unsigned long long cost = fixed_cost();
  if (stats) stats->cost_before = stats->cost_after = lt->cost();
  uint64_t visits_total = 0;
  for (int pass = 0; pass < prm.passes; pass++) {
    const std::vector<uint32_t> par_b = lt->par, kid_b = lt->kid;  // (a pass that raises the cost is undone)
    const std::vector<unsigned long long> bx_b = lt->bx;
    // 1. search (read-only)
    uint64_t visits = 0, found = 0, max_v = 0;
    for (uint32_t x = 0; x < n_nodes; x++) {
      if ((x + (uint32_t)pass) % mu != 0u) { mv[x].y = kNone; continue; }
      mv[x] = reins::find_move(t, x, prm.search);
      visits += mv[x].visits;
      max_v = std::max<uint64_t>(max_v, mv[x].visits);
      found += mv[x].y != kNone;
    }
    visits_total += visits;
    // 2. lock
    std::fill(lock.begin(), lock.end(), 0ull);
    for (uint32_t x = 0; x < n_nodes; x++) {
      if (mv[x].y == kNone) continue;
      const unsigned long long key = reins::move_key(x, mv[x].gain);
      reins::for_move_nodes(t, x, mv[x].y, [&](uint32_t q) { if (lock[q] < key) lock[q] = key; });
    }
    // 3. check: the link locks, then the target's path
    for (uint32_t x = 0; x < n_nodes; x++) {
      holds[x] = 0;
      if (mv[x].y == kNone) continue;
      const unsigned long long key = reins::move_key(x, mv[x].gain);
      bool ok = true;
      reins::for_move_nodes(t, x, mv[x].y, [&](uint32_t q) { if (lock[q] != key) ok = false; });
      holds[x] = ok;
    }
    uint64_t applied = 0, held = 0;
    double gain = 0;
    std::vector<uint8_t> go(n_nodes, 0);  // (decided from `holds` alone, as the device's ri_free_kernel does: not from each other)
    for (uint32_t x = 0; x < n_nodes; x++) {
      if (mv[x].y == kNone) continue;
      held += holds[x];
      const unsigned long long key = reins::move_key(x, mv[x].gain);
      if (holds[x] && reins::target_path_is_free(t, x, mv[x].y, mv[x].lca, [&](uint32_t q) { return holds[q] != 0 && reins::move_key(q, mv[q].gain) > key; })) {
        go[x] = 1; applied++; gain += mv[x].gain;
      }
    }
    // 4. apply
    for (uint32_t x = 0; x < n_nodes; x++)
      if (go[x]) reins::apply_move(t, x, mv[x].y);
    // 5. refit, and the pass's verdict
    refit_links(lt);
    const unsigned long long cost_now = fixed_cost();
    if (stats) { stats->passes++; stats->visits += visits; stats->found += found; stats->max_visits = std::max(stats->max_visits, max_v); }
    if (verbose)
      std::fprintf(stderr, "reinsert pass %2d: %8llu searches found a move, %8llu hold their links, %8llu applied, %.1f visits per search (max %llu), gain %.6g, cost %.6g%s\n", pass,
                   (unsigned long long)found, (unsigned long long)held, (unsigned long long)applied, (double)visits / std::max<uint64_t>(1, (n_nodes + mu - 1) / mu), (unsigned long long)max_v,
                   gain, lt->cost(), cost_now > cost ? " (ROSE: pass undone)" : "");
    if (cost_now > cost) {  // the moves of a pass lock six nodes each, not their paths: their gains need not add up
      lt->par = par_b; lt->kid = kid_b; lt->bx = bx_b;
      t = reins::Tree{n_int, lt->par.data(), lt->kid.data(), lt->bx.data()};
      if (stats) stats->undone = 1;
      break;
    }
    cost = cost_now;
    if (stats) { stats->applied += applied; stats->cost_after = lt->cost(); }
    if (mu == 1 && reins::stop_after_pass(prm.stop, applied, visits_total, n_nodes)) break;
  }
  for (uint32_t i = 0; i < n_int; i++) reins::order_children(t, i);
}

void reinsert_optimize_batch(RefBvh *t, const ReinsertBatchParams &prm, ReinsertBatchStats *stats) {
  if (t->ref_tri.size() < 4) return;
  LinkTree lt;
  link_tree_of(*t, &lt);
  reinsert_batch_links(&lt, prm, stats);
  RefBvh out;
  ref_bvh_of(lt, *t, &out);
  *t = std::move(out);
}

}  // namespace pbrt_hip
